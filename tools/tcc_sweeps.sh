#!/bin/bash
# L2 <-> memory-side counters (TCC_EA0_*) of the two image sweeps of a pass and of the bare loops of tools/hbm_ceiling_bench.bin that
# emulate the albedo sweep (images only / with its side reads and writes), in separate passes of a few counters each (never combined with a
# trace domain other than --kernel-trace).  What separates a 7.1 TB/s read stream from the sweeps' 5.5?   bash tools/tcc_sweeps.sh <dir under gpurun_out>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-r6tcc}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ TCC_EA0_WRREQ_LEVEL" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL" \
           "TCC_TAG_STALL TCC_BUSY GRBM_GUI_ACTIVE TCC_REQ"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/lib$i -- python3 $R/tools/pass_prof.py 2048 4 20 3 > $OUT/lib$i.log 2>&1
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/bare$i -- $R/tools/hbm_ceiling_bench.bin 4096 4096 3 2 > $OUT/bare$i.log 2>&1
done
find $OUT -name "*.db" -delete
python3 - $OUT <<'PY'
import csv,glob,sys,collections,re
out=sys.argv[1]
want=('k_albedo_fused','k_light_fused_mfw','k_sweep_occ')
agg=collections.defaultdict(list); dur=collections.defaultdict(list)
for sub in sorted(glob.glob(f'{out}/lib*/')+glob.glob(f'{out}/bare*/')):
    for f in glob.glob(f'{sub}/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            n=r['Kernel_Name']
            if not any(w in n for w in want): continue
            m=re.search(r'(k_[a-z_]+<[^>]*>)', n); key=m.group(1) if m else n[:60]
            agg[(key, r['Counter_Name'])].append(float(r['Counter_Value']))
            dur[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
med=lambda v: sorted(v)[len(v)//2]
kernels=sorted({k for k,_ in agg})
for k in kernels:
    c={cn: med(v) for (kk,cn),v in agg.items() if kk==k}
    rd, wr = c.get('TCC_EA0_RDREQ',0), c.get('TCC_EA0_WRREQ',0)
    line=f"{k:46s} {med(dur[k]):7.1f} us  RDREQ {rd:.3g} WRREQ {wr:.3g}"
    if rd: line+=f"  rd latency {c.get('TCC_EA0_RDREQ_LEVEL',0)/rd:7.0f} clk"
    if wr: line+=f"  wr latency {c.get('TCC_EA0_WRREQ_LEVEL',0)/wr:7.0f} clk"
    busy=c.get('TCC_BUSY',0)
    for cn in ('TCC_EA0_RDREQ_DRAM_CREDIT_STALL','TCC_EA0_WRREQ_DRAM_CREDIT_STALL','TCC_EA0_WRREQ_STALL','TCC_TOO_MANY_EA_WRREQS_STALL','TCC_TAG_STALL'):
        if cn in c: line+=f"  {cn.replace('TCC_','').replace('EA0_','')} {c[cn]:.3g}"
    if busy: line+=f"  TCC_BUSY {busy:.3g} GUI {c.get('GRBM_GUI_ACTIVE',0):.3g}"
    print(line)
PY
