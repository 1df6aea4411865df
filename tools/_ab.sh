timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o p -- python3 $R/bench.py --steps 5 --no-total-solve > $R/gpurun_out/prof_bench.json 2>/dev/null
cd $R; tail -1 gpurun_out/prof_bench.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','cg_only_us_per_iteration','device_copy_GBs')}); print(d['roofline']); print(d['cpu_baseline'])"
