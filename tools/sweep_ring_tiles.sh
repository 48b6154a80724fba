for T in default 8 16 24 31 32 40 48 61 62; do
  if [ "$T" = default ]; then unset WLSQM_HIP_RING_TILES; else export WLSQM_HIP_RING_TILES=$T; fi
  for i in 1 2; do python3 bench.py --config C5 --steps 20 --warmup 5 --no-parity --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('T=$T', d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
done
