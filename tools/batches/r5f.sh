mkdir -p gpurun_out/r5f; O=gpurun_out/r5f
for i in 1 2 3; do python -m pytest tests/test_gpu_strips.py -x -q -m gpu -k "one_process or recognised" >> $O/strips.log 2>&1; echo rc=$? >> $O/strips.log; done
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo rc=$? >> $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err
grep -E "passed|failed|rc=" $O/strips.log; grep -v "Gloo\|amdgpu\|socket.cpp" $O/pytest.log | tail -8
