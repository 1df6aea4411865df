#!/bin/bash
O=gpurun_out/r5u; mkdir -p $O; rm -f $O/ab.log
python -m pytest tests/test_gpu_edge_and_scale.py -q -m gpu -k "matrix_pipe or tiled_lighting" 2>&1 | tail -2
for rep in 1 2 3; do for r in 1 2 3; do
  echo -n "20 images run$r: " >> $O/ab.log; timeout 300 python3 tools/pass_time.py 2048 4 20 10 light_run=$r 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/ab.log
done; done
for rep in 1 2; do for r in 1 2 3; do
  echo -n "40 images run$r: " >> $O/ab.log; timeout 300 python3 tools/pass_time.py 2048 4 40 6 light_run=$r 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/ab.log
done; done
python3 - <<'PY'
import json,collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r5u/ab.log'):
    n,_,r=l.partition(': ')
    try: j=json.loads(r); d[n].append((j['phase_ms']['energy'], j['ms_per_pass']))
    except Exception as e: d[n].append(str(e)[:40])
for n,v in d.items(): print(n, v)
PY
