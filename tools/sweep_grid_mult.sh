#!/usr/bin/env bash
# usage (GPU box, repo root): bash tools/sweep_grid_mult.sh CONFIG [values...]   — kernel ms / HBM fraction of one BASELINE config per
# WLSQM_HIP_GRID_MULT (workgroups launched per resident workgroup slot of the persistent tile kernels), two passes each
CFG="${1:-C2}"; shift || true
VALS=("$@"); if (( ${#VALS[@]} == 0 )); then VALS=(default 4 8 12 16 24 32 64); fi
for M in "${VALS[@]}"; do
  if [ "$M" = default ]; then unset WLSQM_HIP_GRID_MULT; else export WLSQM_HIP_GRID_MULT=$M; fi
  for i in 1 2; do python3 bench.py --config "$CFG" --steps 20 --warmup 5 --no-parity --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('GRID_MULT=$M', d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
done
