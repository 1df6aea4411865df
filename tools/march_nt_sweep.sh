#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
for size in 2304 2560 2816 3072 3584 4096; do
  for o in "march_nt=0" "march_nt=1"; do
    echo -n "$size $o: "; timeout 300 python3 tools/cg_time.py $size 4 full cg_resident=0 $o 2>/dev/null | grep "^{" | cut -c1-120
  done
done
for o in "march_nt=0" "march_nt=1"; do echo -n "4096 sf2 $o: "; timeout 300 python3 tools/cg_time.py 4096 2 full $o 2>/dev/null | grep "^{" | cut -c1-120; done
