#!/bin/bash
# Same-box A/B of library variants on the whole pass (tools/pass_time.py): every srmeetsps-cuda_amd/variants/*.so in turn, three rounds
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
cp srmeetsps-cuda_amd/libsrps_hip.so /tmp/keep.so
for rep in 1 2 3; do
  for f in ${VARIANTS:-srmeetsps-cuda_amd/variants}/*.so; do
    cp "$f" srmeetsps-cuda_amd/libsrps_hip.so
    echo -n "$(basename "$f" .so): "
    timeout 300 python3 tools/pass_time.py 2048 4 20 10 "$@" 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
cp /tmp/keep.so srmeetsps-cuda_amd/libsrps_hip.so
