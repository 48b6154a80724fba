timeout -k 10 600 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "operator_path or capture_into" 2>&1 | tail -2
for i in 1 2 3; do python bench.py --config C4 --steps 5 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4', d['ms_per_step'], d['roofline']['frac'])"; done
python tools/time_solve_many.py 2>&1 | tail -12
