mkdir -p gpurun_out/r5e; O=gpurun_out/r5e
bash tools/ab_variants.sh "2048 4 full" > $O/ab.log 2>&1
VARIANTS=srmeetsps-cuda_amd/variants_l bash tools/ab_pass.sh > $O/ab_light.log 2>&1
python -m pytest tests/test_gpu_strips.py tests/test_gpu_edge_and_scale.py tests/test_gpu_full_size.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo rc=$? >> $O/pytest.log
cat $O/ab.log; cut -c1-250 $O/ab_light.log; grep -v "Gloo\|amdgpu\|socket.cpp" $O/pytest.log | tail -12
