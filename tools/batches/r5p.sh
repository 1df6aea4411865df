#!/bin/bash
O=gpurun_out/r5p; mkdir -p $O
python -m pytest tests/test_gpu_edge_and_scale.py -q -m gpu -k "matrix_pipe" -s > $O/test.log 2>&1
grep -E "passed|failed|Error|error" $O/test.log | cut -c1-250
for rep in 1 2 3; do for r in 1 2 3; do
  timeout 300 python3 tools/pass_time.py 2048 4 20 10 light_run=$r 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/ab.log
done; done
cut -c1-215 $O/ab.log
