#!/bin/bash
# round 4, first GPU call: new tests, the bench line, PMC on the image sweeps, the HBM ceiling with the guide's recipe swept
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04a
mkdir -p "$OUT"
cd "$R"
timeout 1800 python -m pytest tests -m gpu -x -q > "$OUT/pytest_new.log" 2>&1
tail -5 "$OUT/pytest_new.log"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -c 600 "$OUT/bench.json"
timeout 300 tools/hbm_ceiling_bench.bin 4096 4096 10 1 > "$OUT/hbm_ceiling.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
for mode in 0 2; do
  for pmc in FETCH_SIZE WRITE_SIZE; do
    sub=$( [ $pmc = FETCH_SIZE ] && echo fetch || echo write )
    timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$OUT/pmc_sweeps_mode$mode/$sub" -- python3 "$R/tools/pass_prof.py" 2048 4 20 3 albedo_mode=$mode > "$OUT/pmc_sweeps_mode${mode}_$sub.log" 2>&1
  done
  python3 "$R/tools/pmc_sweeps.py" "$OUT/pmc_sweeps_mode$mode" 4194304 20 3 "$OUT/sweeps_mode$mode.json" > /dev/null 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/tools/pass_prof.py" 2048 4 20 6 > "$OUT/stats.log" 2>&1
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
ls "$OUT"
