timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pass2 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --no-cpu-baseline --no-total-solve > $GRAFT_REPO_ROOT/gpurun_out/pass2.json 2>/dev/null
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/pass2.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', round(d['value']), 'ms', round(d['ms_per_step'],3), 'cg_only us/it', round(d['cg_only_us_per_iteration'],2))"
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/pass2/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_light_solve' in r['Kernel_Name']]
a,b=idx[-3],idx[-2]
t0=int(rows[a]['Start_Timestamp']); tot=0; prev_end=None
print("pass wall: %.1f us"%((int(rows[b]['Start_Timestamp'])-t0)/1e3))
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=(s-prev_end)/1e3 if prev_end else 0
    prev_end=e; tot+=(e-s)
    print("%-46s %8.1f us  gap %6.1f"%(r['Kernel_Name'][:46],(e-s)/1e3,gap))
print("sum kernels %.1f us"%(tot/1e3))
PY
