#!/bin/bash
# Same-box A/B/C... of several builds of the library (development aid): every srmeetsps-cuda_amd/variants/*.so takes the place of
# libsrps_hip.so in turn, three rounds, microseconds per CG step of tools/cg_time.py (default: the resident CG at 2048 x 2048, sf 4, full mask).
#   gpurun -- 'bash tools/ab_variants.sh "2048 4 full"'
ARGS=${1:-"2048 4 full"}
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
cp srmeetsps-cuda_amd/libsrps_hip.so /tmp/keep.so
for rep in 1 2 3; do
  for f in ${VARIANTS:-srmeetsps-cuda_amd/variants}/*.so; do
    cp "$f" srmeetsps-cuda_amd/libsrps_hip.so
    echo -n "$(basename "$f" .so): "
    timeout 300 python3 tools/cg_time.py $ARGS 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
cp /tmp/keep.so srmeetsps-cuda_amd/libsrps_hip.so
