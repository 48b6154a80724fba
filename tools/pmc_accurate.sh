#!/usr/bin/env bash
# usage (GPU box, repo root): bash tools/pmc_accurate.sh [CONFIGS] [KERNEL_SUBSTRING]  — three SQ counter passes (rocprofv3 --pmc with
# --kernel-trace only) of the accurate mode's kernel on 1M cases of the given configs (tools/time_accurate.py); per-kernel means
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
CFGS="${1:-C2}"; KSUB="${2:-fit_accurate_kernel}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
export WLSQM_TIME_MODES="${WLSQM_TIME_MODES:-accurate}"
for grp in "sq1:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "sq2:SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "sq3:SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM"; do
  tag=${grp%%:*}; ctr=${grp#*:}
  echo "### $CFGS, kernels matching '$KSUB': $tag"
  timeout -k 10 300 bash tools/pmc_cmd.sh acc_$tag "$ctr" "$KSUB" tools/time_accurate.py 1000000 "$CFGS" 2>&1 | grep -v "^$" | tail -30
done
