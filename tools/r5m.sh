mkdir -p gpurun_out/r5m; O=gpurun_out/r5m
bash tools/ab_pass.sh > $O/ab_pass.log 2>&1
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo rc=$? >> $O/pytest.log
timeout 200 python tools/stress_resident.py 60 > $O/stress.log 2>&1; echo rc=$? >> $O/stress.log
cut -c1-200 $O/ab_pass.log; grep -v "Gloo\|amdgpu\|socket.cpp" $O/pytest.log | tail -6; tail -4 $O/stress.log
