#!/bin/bash
# variants of the decoupled-wave matrix-pipe sweep (light_run = 3), same box, three rounds; the vector form (light_run = 1) of the first variant beside them
O=gpurun_out/r5q2; mkdir -p $O; rm -f $O/ab.log
cp srmeetsps-cuda_amd/libsrps_hip.so /tmp/keep.so
for rep in 1 2 3; do
  echo -n "vector: " >> $O/ab.log
  timeout 300 python3 tools/pass_time.py 2048 4 20 10 light_run=1 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/ab.log
  for f in srmeetsps-cuda_amd/variants_m/*.so; do
    cp "$f" srmeetsps-cuda_amd/libsrps_hip.so
    echo -n "$(basename "$f" .so): " >> $O/ab.log
    timeout 300 python3 tools/pass_time.py 2048 4 20 10 light_run=3 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/ab.log
  done
  cp /tmp/keep.so srmeetsps-cuda_amd/libsrps_hip.so
done
python3 - <<'PY'
import json,collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r5q2/ab.log'):
    n,_,r=l.partition(': ')
    try: d[n].append(json.loads(r)['phase_ms']['energy'])
    except Exception as e: d[n].append(str(e)[:40])
for n,v in d.items(): print(n, v)
PY
