#!/bin/bash
# image loads in flight per thread of the albedo sweep (SRPS_ALBEDO_UNROLL 1 2 3 4 6), variants in turn on one box, three rounds
O=gpurun_out/r5v; mkdir -p $O; rm -f $O/ab.log
cp srmeetsps-cuda_amd/libsrps_hip.so /tmp/keep.so
for rep in 1 2 3; do
  for f in srmeetsps-cuda_amd/variants_m/a_u*.so; do
    cp "$f" srmeetsps-cuda_amd/libsrps_hip.so
    echo -n "$(basename "$f" .so): " >> $O/ab.log
    timeout 300 python3 tools/pass_time.py 2048 4 20 10 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/ab.log
  done
  cp /tmp/keep.so srmeetsps-cuda_amd/libsrps_hip.so
done
python3 - <<'PY'
import json,collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r5v/ab.log'):
    n,_,r=l.partition(': ')
    try: j=json.loads(r); d[n].append((j['phase_ms']['albedo_sweep'], j['ms_per_pass']))
    except Exception as e: d[n].append(str(e)[:40])
for n,v in d.items(): print(n, v)
PY
