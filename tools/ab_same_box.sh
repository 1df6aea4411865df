#!/bin/bash
# Same-box A/B of two builds of the library (development aid).  The boxes of the pool differ by up to 20 % for the same binary,
# so a change is only judged against the build it replaces ON THE SAME BOX, alternately:
#   make -C srmeetsps-cuda_amd/csrc                      # base  -> srmeetsps-cuda_amd/libsrps_hip.so
#   cp srmeetsps-cuda_amd/libsrps_hip.so /tmp/base.so; <edit>; make ...; cp srmeetsps-cuda_amd/libsrps_hip.so srmeetsps-cuda_amd/libsrps_hip_exp.so
#   cp /tmp/base.so srmeetsps-cuda_amd/libsrps_hip.so
#   gpurun -- 'bash tools/ab_same_box.sh "2048 4 1 0 101 1"'      # arguments of tools/cg_prof.py: size sf recompute strip steps resident
# Prints microseconds per CG step, three rounds of base / exp.  (libsrps_hip_exp.so is not tracked; delete it afterwards.)
ARGS=${1:-"2048 4 1 0 101 1"}
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
[ -f srmeetsps-cuda_amd/libsrps_hip_exp.so ] || { echo "no srmeetsps-cuda_amd/libsrps_hip_exp.so"; exit 1; }
cp srmeetsps-cuda_amd/libsrps_hip.so /tmp/base.so
for rep in 1 2 3; do
  for v in base exp; do
    if [ $v = exp ]; then cp srmeetsps-cuda_amd/libsrps_hip_exp.so srmeetsps-cuda_amd/libsrps_hip.so; else cp /tmp/base.so srmeetsps-cuda_amd/libsrps_hip.so; fi
    echo -n "$v: "
    timeout 300 python3 tools/cg_prof.py $ARGS 2>&1 | grep -v amdgpu.ids | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print(round(d['seconds']*1e6/d['iterations'],3), 'us per step')"
  done
done
cp /tmp/base.so srmeetsps-cuda_amd/libsrps_hip.so
