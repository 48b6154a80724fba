#!/usr/bin/env bash
# usage (GPU box, repo root): bash tools/pmc_new_kernels.sh   — three SQ counter passes (rocprofv3 --pmc with --kernel-trace only) of the
# index-based ring kernel (tools/time_cloud.py) and of the strict row-per-lane kernel (tools/time_strict.py); per-kernel means
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for grp in "sq1:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "sq2:SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "sq3:SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM"; do
  tag=${grp%%:*}; ctr=${grp#*:}
  echo "### gather ring (fit_ring_kernel<2,4,64,16,1,true>), 1M index-based C3 cases: $tag"
  timeout -k 10 200 bash tools/pmc_cmd.sh gring_$tag "$ctr" "fit_ring_kernel<2, 4, 64, 16, 1, true>" tools/time_cloud.py 1000000 2>&1 | grep -v "^$" | tail -9
  echo "### strict row kernel (fit_strict_rows_kernel<2,4,16>), 200k C3 cases: $tag"
  timeout -k 10 200 bash tools/pmc_cmd.sh srows_$tag "$ctr" "fit_strict_rows_kernel<2, 4" tools/time_strict.py 200000 2>&1 | grep -v "^$" | tail -9
done
