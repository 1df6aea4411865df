#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-r04f}; mkdir -p $OUT      # bash tools/r04_sq.sh <output directory under gpurun_out>
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq1 -- python3 $R/tools/pass_prof.py 2048 4 20 3 > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- python3 $R/tools/pass_prof.py 2048 4 20 3 > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq3 -- python3 $R/tools/pass_prof.py 2048 4 20 3 > $OUT/sq3.log 2>&1
find $OUT -name "*.db" -delete
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
for sub in ('sq1','sq2','sq3'):
    fs=glob.glob(f'{out}/{sub}/**/*counter_collection.csv',recursive=True)
    if not fs: print(sub,'no csv'); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        n=r['Kernel_Name']
        if 'k_light_fused_tile' in n or 'k_light_fused_mf' in n or 'k_albedo_fused' in n:
            agg[(n.split('(')[0][-40:], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print(sub, k, sorted(v)[len(v)//2])
PY
