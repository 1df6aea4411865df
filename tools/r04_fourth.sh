#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r04d
mkdir -p "$OUT"
cd "$R"
timeout 2700 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1
tail -15 "$OUT/pytest.log"
grep -h "Mitten:" "$OUT/pytest.log" | head
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; python3 - "$OUT/bench.json" <<'PY'
import json,sys
b=json.load(open(sys.argv[1]))
print(b['value'], b['ms_per_step'], b['roofline']['frac'], b['cg_only_us_per_iteration'])
for k,v in b['legs'].items():
    print(k, {kk: vv for kk,vv in v.items() if kk in ('cg_only_us_per_iteration','ms_per_step','us_per_step_and_rank','total_solve_s')}, v.get('roofline',{}).get('frac'))
PY
