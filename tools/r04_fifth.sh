#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04e
mkdir -p "$OUT"
cd "$R"
timeout 2700 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1
grep -E "passed|failed|^FAILED" "$OUT/pytest.log" | tail -12
timeout 600 python -m pytest tests/test_mitten_full.py -m gpu -q -s 2>&1 | grep "Mitten" | cut -c1-600
for rep in 1 2; do
  for o in "image_tiles=0" "image_tiles=1" "image_tiles=1 albedo_mode=0" "image_tiles=0 albedo_mode=0"; do
    timeout 300 python3 tools/pass_time.py 2048 4 20 8 $o 2>&1 | grep '^{' >> "$OUT/pass_tiles.jsonl"
  done
done
cat "$OUT/pass_tiles.jsonl"
SRPS_BYTES=1 timeout 300 python3 tools/pass_time.py 2048 4 20 8 2>&1 | grep '^{'
timeout 300 python3 tools/pass_time.py 4096 2 16 3 2>&1 | grep '^{'
timeout 300 python3 tools/pass_time.py 4096 2 16 3 image_tiles=0 2>&1 | grep '^{'
timeout 300 python3 tools/pass_time.py 1024 4 20 8 2>&1 | grep '^{'
