#!/usr/bin/env bash
# usage (GPU box, repo root): bash tools/pmc_refine.sh [NCASES DIM,ORDER,K "KERNEL SUBSTRING"]  — SQ passes of the one-lane-per-case refinement
# kernel (default: configs[2]'s shape, 400k cases, 64 neighbours; max_iter 0 / 1 / 2 / 4 / 10 averaged) and kernel stats of the same command.
set -uo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
export TIME_REFINE_MODES=stage
N="${1:-400000}"; SHAPE="${2:-2,4,64}"; KSUB="${3:-fit_stage_refine_kernel<2, 4}"
for grp in "sq1:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "sq2:SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "sq3:SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" "mem:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag="${grp%%:*}"; ctr="${grp#*:}"
  echo "== $tag"
  timeout -k 10 200 bash tools/pmc_cmd.sh refine_$tag "$ctr" "$KSUB" tools/time_refine.py $N $SHAPE 2>&1 | grep -v "^$" | tail -9
done
echo "== kernel stats"
timeout -k 10 200 bash tools/prof_cmd.sh refine_stats tools/time_refine.py $N $SHAPE 2>&1 | tail -6
