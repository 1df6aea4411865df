#!/bin/bash
# same-box A/B: libsrps_hip.so (base) against libsrps_hip_exp.so
cd $GRAFT_REPO_ROOT
cp srmeetsps-cuda_amd/libsrps_hip.so /tmp/base.so
for rep in 1 2 3; do
  for v in base exp; do
    if [ $v = exp ]; then cp srmeetsps-cuda_amd/libsrps_hip_exp.so srmeetsps-cuda_amd/libsrps_hip.so; else cp /tmp/base.so srmeetsps-cuda_amd/libsrps_hip.so; fi
    echo -n "$v: "; timeout 300 python tools/cg_prof.py 2048 4 1 0 101 1 2>&1 | grep -v amdgpu.ids | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print(round(d[\"seconds\"]*1e6/d[\"iterations\"],3), \"us per step\")"
  done
done
cp /tmp/base.so srmeetsps-cuda_amd/libsrps_hip.so
