#!/bin/bash
# The round-end check on a GPU box: the whole GPU suite, smoke(), the bench line.   gpurun --timeout 2700 -- bash tools/gpu_check.sh
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/check
mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1
grep -E "passed|failed|^FAILED" "$OUT/pytest.log" | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; python3 - "$OUT/bench.json" <<'PY'
import json,sys
b=json.load(open(sys.argv[1]))
print(b['value'], b['ms_per_step'], b['roofline']['frac'], b['cg_only_us_per_iteration'], b['total_solve_s'], b['total_solve_with_setup_s'])
for k,v in b['legs'].items():
    print(k, {kk: vv for kk,vv in v.items() if kk in ('cg_only_us_per_iteration','ms_per_step','us_per_step_and_rank','total_solve_s','best_read_GBs')}, v.get('roofline',{}).get('frac'))
PY
