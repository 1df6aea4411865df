#!/bin/bash
# as ab_variants.sh, over several argument sets:  bash tools/ab_variants2.sh "2048 4 ellipse" "2048 4 full" ...
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
cp srmeetsps-cuda_amd/libsrps_hip.so /tmp/keep.so
for rep in 1 2 3; do
  for ARGS in "$@"; do
    for f in srmeetsps-cuda_amd/variants/*.so; do
      cp "$f" srmeetsps-cuda_amd/libsrps_hip.so
      echo -n "[$ARGS] $(basename "$f" .so): "
      timeout 300 python3 tools/cg_time.py $ARGS 2>&1 | grep -v amdgpu.ids | tail -1
    done
  done
done
cp /tmp/keep.so srmeetsps-cuda_amd/libsrps_hip.so
