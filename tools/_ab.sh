cd /tmp && export TMPDIR=/tmp
for o in 1024 0 512 1536; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/lb$o -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --no-cpu-baseline --no-total-solve --option light_blocks=$o > $GRAFT_REPO_ROOT/gpurun_out/lb$o.json 2>/dev/null
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,json
for o in (1024,0,512,1536):
    d=json.loads(open("gpurun_out/lb%d.json"%o).read().strip().split('\n')[-1])
    print("light_blocks",o,"value",round(d['value']),"ms",round(d['ms_per_step'],3))
    f=glob.glob("gpurun_out/lb%d/**/*kernel_stats.csv"%o,recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if "k_light" in r["Name"]: print("   ",r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
