#!/bin/bash
# round 4, second GPU call: the full GPU suite on the new build, then same-box A/B of the cache policies and the tiled lighting sweep
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04b
mkdir -p "$OUT"
cd "$R"
timeout 1800 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
tail -5 "$OUT/pytest.log"
V=$R/srmeetsps-cuda_amd/variants
for rep in 1 2; do
  for lib in "" $V/nt0.so $V/nt1s1.so; do
    for o in "light_tiled=0" "light_tiled=1" "light_tiled=1 albedo_mode=2"; do
      SRPS_LIB_PATH=$lib timeout 300 python3 tools/pass_time.py 2048 4 20 8 $o 2>&1 | grep '^{' >> "$OUT/pass_ab.jsonl"
    done
  done
done
for rep in 1 2; do
  for o in "march_nt=0" "march_nt=1"; do
    timeout 300 python3 tools/cg_time.py 4096 2 full $o 2>&1 | grep '^{' >> "$OUT/cg4096_ab.txt"
    timeout 300 python3 tools/cg_time.py 2048 4 full cg_resident=0 $o 2>&1 | grep '^{' >> "$OUT/cg2048s_ab.txt"
  done
done
SRPS_BYTES=1 timeout 300 python3 tools/pass_time.py 2048 4 20 8 2>&1 | grep '^{' >> "$OUT/pass_bytes.jsonl"
SRPS_BYTES=1 SRPS_LIB_PATH=$V/nt0.so timeout 300 python3 tools/pass_time.py 2048 4 20 8 light_tiled=0 2>&1 | grep '^{' >> "$OUT/pass_bytes.jsonl"
timeout 300 python3 tools/pass_time.py 4096 2 16 3 2>&1 | grep '^{' >> "$OUT/pass_4096.jsonl"
SRPS_LIB_PATH=$V/nt0.so timeout 300 python3 tools/pass_time.py 4096 2 16 3 light_tiled=0 march_nt=0 2>&1 | grep '^{' >> "$OUT/pass_4096.jsonl"
cat "$OUT/pass_ab.jsonl" "$OUT/cg4096_ab.txt" "$OUT/cg2048s_ab.txt" "$OUT/pass_bytes.jsonl" "$OUT/pass_4096.jsonl"
