#!/usr/bin/env bash
# usage (GPU box, repo root): bash tools/sweep_ring_tiles.sh CONFIG [values...]   — kernel ms / HBM fraction of C3 or C5 per
# WLSQM_HIP_RING_TILES (tiles per workgroup of the ring kernel; `default` = the launch's own choice), two passes each
CFG="${1:-C5}"; shift || true
VALS=("$@"); if (( ${#VALS[@]} == 0 )); then VALS=(default 4 8 12 16 24 31 32 40 48 61 62); fi
for T in "${VALS[@]}"; do
  if [ "$T" = default ]; then unset WLSQM_HIP_RING_TILES; else export WLSQM_HIP_RING_TILES=$T; fi
  for i in 1 2; do python3 bench.py --config "$CFG" --steps 20 --warmup 5 --no-parity --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$CFG T=$T', d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
done
