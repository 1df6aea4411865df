#!/bin/bash
# round 5, after the lighting sweep moved to the matrix pipe: kernel statistics of the bench command, counter bytes and SQ counters of the sweeps
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/prof_sweeps; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -- python3 "$R/bench.py" --steps 5 --no-cpu-baseline --no-total-solve > "$OUT/bench.json" 2> "$OUT/bench.err"
for pmc in FETCH_SIZE WRITE_SIZE; do
  sub=$( [ $pmc = FETCH_SIZE ] && echo fetch || echo write )
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$OUT/pmc_sweeps/$sub" -- python3 "$R/tools/pass_prof.py" 2048 4 20 3 > "$OUT/pmc_sweeps_$sub.log" 2>&1
done
python3 "$R/tools/pmc_sweeps.py" "$OUT/pmc_sweeps" 4194304 20 3 "$OUT/sweeps_default.json" > /dev/null 2>&1
cd $R; bash tools/sq_sweeps.sh prof_sweeps/sq > $OUT/sq_sweeps.txt 2>&1
python3 bench.py > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
ls $OUT $OUT/bench/*; cat $OUT/sweeps_default.json | head -60; cat $OUT/sq_sweeps.txt | tail -30
