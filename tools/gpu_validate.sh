#!/bin/bash
O=gpurun_out/validate; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo rc=$? >> $O/pytest.log
grep -v "Gloo\|amdgpu\|socket.cpp" $O/pytest.log | tail -8
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo smoke rc=$?; tail -2 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err; echo bench rc=$?
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/validate/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','total_solve_s','total_solve_with_setup_s','setup_s')}, d['roofline']['frac'])
PY
