#!/bin/bash
# light_run = 2 (the lighting contraction on the matrix pipe) against light_run = 1: parity test, then alternating passes on one box
O=gpurun_out/r5m; mkdir -p $O
python -m pytest tests/test_gpu_edge_and_scale.py -x -q -m gpu -k "matrix_pipe" -s > $O/test.log 2>&1
tail -15 $O/test.log
for i in 1 2 3; do
python tools/pass_time.py 2048 4 20 10 light_run=1 >> $O/pass.jsonl 2>>$O/pass.err
python tools/pass_time.py 2048 4 20 10 light_run=2 >> $O/pass.jsonl 2>>$O/pass.err
done
cut -c1-260 $O/pass.jsonl
