mkdir -p gpurun_out/r5n; O=gpurun_out/r5n
python -m pytest tests -x -q -m gpu > $O/pytest1.log 2>&1; echo rc=$? >> $O/pytest1.log
python -m pytest tests -x -q -m gpu -s -k "strips or comm or distributed or one_process" > $O/pytest2.log 2>&1; echo rc=$? >> $O/pytest2.log
python bench.py > $O/bench.json 2> $O/bench.err
grep -v "Gloo\|amdgpu\|socket.cpp" $O/pytest1.log | tail -4; grep -E "passed|failed|rc=|attempts" $O/pytest2.log | tail -6
