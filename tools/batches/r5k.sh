mkdir -p gpurun_out/r5k; O=gpurun_out/r5k
bash tools/ab_variants.sh "2048 4 full" > $O/ab.log 2>&1
bash tools/ab_variants.sh "2048 4 ellipse" > $O/ab_ellipse.log 2>&1
python -m pytest tests/test_gpu_full_size.py tests/test_gpu_strips.py tests/test_gpu_edge_and_scale.py -x -q -m gpu -k "metric_size or bit_for_bit or one_wait or tile" > $O/pytest.log 2>&1; echo rc=$? >> $O/pytest.log
cat $O/ab.log $O/ab_ellipse.log; grep -v "Gloo\|amdgpu\|socket.cpp" $O/pytest.log | tail -5
