#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04c
mkdir -p "$OUT"
cd "$R"
timeout 2400 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log"
timeout 600 tools/hbm_ceiling_bench.bin 4096 4096 10 1 > "$OUT/hbm_ceiling.txt" 2>&1
grep -E "march_|sweep_" "$OUT/hbm_ceiling.txt"
timeout 1200 python3 tools/albedo_mode_compare.py all > "$OUT/albedo_modes.jsonl" 2> "$OUT/albedo_modes.err"
cat "$OUT/albedo_modes.jsonl"
for rep in 1 2 3; do
  for o in "march_nt=0" "march_nt=1"; do
    timeout 300 python3 tools/cg_time.py 4096 2 full $o 2>&1 | grep '^{' >> "$OUT/cg4096_ab.txt"
  done
done
cat "$OUT/cg4096_ab.txt"
timeout 300 python3 tools/pass_time.py 2048 4 20 8 2>&1 | grep '^{' >> "$OUT/pass.jsonl"
timeout 300 python3 tools/pass_time.py 2048 4 20 8 albedo_mode=2 2>&1 | grep '^{' >> "$OUT/pass.jsonl"
cat "$OUT/pass.jsonl"
