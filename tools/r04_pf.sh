#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
V=srmeetsps-cuda_amd/variants
for rep in 1 2; do
  for lib in "" $V/*.so; do
    SRPS_LIB_PATH=$lib timeout 300 python3 tools/pass_time.py 2048 4 20 8 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'] or 'default', d['ms_per_pass'], d['phase_ms'])"
  done
done
