timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', round(d['value']), 'ms', round(d['ms_per_step'],3), 'cg_only us/it', round(d['cg_only_us_per_iteration'],2), 'solve', round(d['total_solve_s'],4))"
