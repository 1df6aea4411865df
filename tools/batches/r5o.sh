#!/bin/bash
O=gpurun_out/r5o; mkdir -p $O
cp srmeetsps-cuda_amd/libsrps_hip.so /tmp/keep.so
for rep in 1 2 3; do
  for f in srmeetsps-cuda_amd/variants_m/*.so; do
    cp "$f" srmeetsps-cuda_amd/libsrps_hip.so
    n=$(basename "$f" .so); opt=light_run=2; [ "$n" = a_vector ] && opt=light_run=1
    echo -n "$n: " >> $O/ab.log
    timeout 300 python3 tools/pass_time.py 2048 4 20 10 $opt 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/ab.log
  done
done
cp /tmp/keep.so srmeetsps-cuda_amd/libsrps_hip.so
cut -c1-215 $O/ab.log
