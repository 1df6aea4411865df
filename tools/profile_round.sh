#!/bin/bash
# Profiles of one round, run on the GPU box:  bash tools/profile_round.sh r02
# kernel statistics of the bench command (rocprofv3 --kernel-trace --stats) and the PMC passes the roofline's `traffic`
# comes from (FETCH_SIZE, WRITE_SIZE in separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; never combined
# with a trace domain other than --kernel-trace), plus SQ instruction counters of the resident kernel.  Raw output goes to
# gpurun_out/prof_<round>/; tools/pmc_traffic.py turns it into the files under profiles/.
set -u
ROUND=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -- python3 "$R/bench.py" --steps 5 --no-cpu-baseline --no-total-solve --no-live-traffic > "$OUT/bench.json" 2> "$OUT/bench.err"
for cfg in "2048 4 1 0 101 1 resident_2048" "2048 4 1 0 101 0 streaming_2048" "4096 2 1 0 101 0 streaming_4096"; do
    set -- $cfg
    for pmc in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$OUT/pmc_${7}_$pmc" -- python3 "$R/tools/cg_prof.py" $1 $2 $3 $4 $5 $6 > "$OUT/pmc_${7}_$pmc.log" 2>&1
    done
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/pmc_resident_2048_SQ1" -- python3 "$R/tools/cg_prof.py" 2048 4 1 0 101 1 > "$OUT/pmc_resident_2048_SQ1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_resident_2048_SQ2" -- python3 "$R/tools/cg_prof.py" 2048 4 1 0 101 1 > "$OUT/pmc_resident_2048_SQ2.log" 2>&1
python3 "$R/tools/valu_issue.py" "$OUT/valu_issue.json" > "$OUT/valu_issue.log" 2>&1
# round 4: the image sweeps -- FETCH_SIZE / WRITE_SIZE (separate passes) of whole passes at the metric's configuration, default options
# and with the reference's albedo CG (albedo_mode=0: num / den / image sums written, read back by the CG and the assembly)
for mode in 3 0; do
  for pmc in FETCH_SIZE WRITE_SIZE; do
    sub=$( [ $pmc = FETCH_SIZE ] && echo fetch || echo write )
    rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$OUT/pmc_sweeps_mode$mode/$sub" -- python3 "$R/tools/pass_prof.py" 2048 4 20 3 albedo_mode=$mode > "$OUT/pmc_sweeps_mode${mode}_$sub.log" 2>&1
  done
  python3 "$R/tools/pmc_sweeps.py" "$OUT/pmc_sweeps_mode$mode" 4194304 20 3 "$OUT/sweeps_mode$mode.json" > /dev/null 2>&1
done
python3 "$R/bench.py" > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err"
"$R/tools/hbm_ceiling_bench.bin" 4096 4096 10 1 > "$OUT/hbm_ceiling.txt" 2>&1
# the occupied-tiles curve and the cache-policy sweep of the streaming step (round 4; kernels unchanged since): only when asked for
if [ "${SRPS_PROFILE_LONG:-0}" = 1 ]; then
python3 "$R/tools/cliff_curve.py" 2>/dev/null | grep "^{" > "$OUT/cliff_curve.jsonl"
fi
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
ls "$OUT"
