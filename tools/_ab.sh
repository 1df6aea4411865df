timeout 300 python tools/_res_check.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --steps 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', round(d['value']), 'ms', round(d['ms_per_step'],3), 'cg_only us/it', round(d['cg_only_us_per_iteration'],2), 'solve', round(d['total_solve_s'],4), d['energies'][-1])"; done
