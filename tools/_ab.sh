timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python bench.py --steps 10 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_res.json; python -c "
import json; d=json.load(open('gpurun_out/bench_res.json')); print({k:d[k] for k in ('value','ms_per_step','cg_only_us_per_iteration','total_solve_s')}); print(d['roofline'])"
