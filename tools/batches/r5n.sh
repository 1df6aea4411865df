#!/bin/bash
O=gpurun_out/r5n; mkdir -p $O
python -m pytest tests/test_gpu_edge_and_scale.py -q -m gpu -k "matrix_pipe" -s > $O/test.log 2>&1
grep -E "passed|failed|energies" $O/test.log | cut -c1-250
VARIANTS=srmeetsps-cuda_amd/variants_m bash tools/ab_pass.sh light_run=2 > $O/ab.log 2>&1
cut -c1-200 $O/ab.log
