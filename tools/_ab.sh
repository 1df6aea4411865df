timeout 300 python tools/_res_check.py 2>&1 | grep -v amdgpu.ids
for o in "cg_resident=1" "cg_resident_debug=1"; do timeout 300 python bench.py --steps 5 --no-cpu-baseline --no-total-solve --option $o 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$o', 'value', round(d['value']), 'ms', round(d['ms_per_step'],3), 'cg_only us/it', round(d['cg_only_us_per_iteration'],2))"; done
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
