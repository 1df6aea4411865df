mkdir -p gpurun_out/r5h; O=gpurun_out/r5h
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo rc=$? >> $O/pytest.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo rc=$? >> $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py > $O/bench2.json 2> $O/bench2.err
grep -v "Gloo\|amdgpu\|socket.cpp" $O/pytest.log | tail -6; tail -2 $O/smoke.log
