cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o p -- python3 $R/bench.py --steps 5 --no-total-solve > $R/gpurun_out/prof_bench.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o p -- python3 $R/tools/cg_prof.py 2048 4 1 0 101 1 > /dev/null 2>&1
done
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $R/gpurun_out/pmc_TCC -o p -- python3 $R/tools/cg_prof.py 2048 4 1 0 101 1 > /dev/null 2>&1
cd $R; tail -1 gpurun_out/prof_bench.json | cut -c1-300
for o in "cg_resident_debug=1"; do timeout 300 python bench.py --steps 3 --no-cpu-baseline --no-total-solve --option $o 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$o', 'cg_only us/it', round(d['cg_only_us_per_iteration'],2))"; done
