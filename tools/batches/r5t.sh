#!/bin/bash
# byte images: the sweeps of the vector form (light_run = 1) and of the matrix form (3), depth of the byte pipeline
O=gpurun_out/r5t; mkdir -p $O; rm -f $O/ab.log
export SRPS_BYTES=1
cp srmeetsps-cuda_amd/libsrps_hip.so /tmp/keep.so
for rep in 1 2 3; do
  for r in 1 2; do echo -n "run$r: " >> $O/ab.log; timeout 300 python3 tools/pass_time.py 2048 4 20 10 light_run=$r 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/ab.log; done
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
for l in open('gpurun_out/r5t/ab.log'):
    n,_,r=l.partition(': ')
    try: j=json.loads(r); d[n].append((j['phase_ms']['energy'], j['ms_per_pass']))
    except Exception as e: d[n].append(str(e)[:40])
for n,v in d.items(): print(n, v)
PY
