#!/bin/bash
# First contact with a multi-GPU node (round 6; nothing of this repo has ever run on two devices: DESIGN.md section 7).
#     bash tools/multi_gpu_first_contact.sh [N=8]
# Runs, in this order and each as a fresh child process, and stops at the first step that fails, naming it:
#   1. pytest tests/test_multi_gpu.py -x          two-rank RCCL runs of srps_execute_sharded (images, overlap_exchange, strips), srps --gpus 2
#   2. bench.py --gpus 2                          BASELINE.json configs[3] on two ranks (library communicator; falls back to torch collectives by itself)
#   3. bench.py --gpus N                          ... on all N
#   4. bench.py --gpus N --config 5               configs[4]: 4096 x 4096, sf 2, 64 images, the depth CG on strips (resident -> streaming -> replicated)
#   5. the same with every fall-back forced (SRPS_FORCE_FAIL): the degrade chain on the real transport
# Every bench line goes to gpurun_out/first_contact/; `config.parallelism` and `config.degraded` say which form ran.
# Predictions to check the first numbers against: DESIGN.md section 7, "What the first scaling curve should look like".
set -u
N=${1:-8}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/first_contact
mkdir -p "$OUT"
cd "$R" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
have=$(python3 -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
echo "devices visible: $have (asked for $N)"
if [ "$have" -lt 2 ]; then echo "FIRST CONTACT: needs at least two devices; nothing run"; exit 3; fi
[ "$have" -lt "$N" ] && N=$have
step() {      # step <name> <log> <command ...>
    local name=$1 log=$2; shift 2
    echo "=== $name"
    if ! timeout 3000 "$@" > "$OUT/$log" 2> "$OUT/${log%.*}.err"; then
        echo "FIRST CONTACT FAILED AT: $name   (output: $OUT/$log, $OUT/${log%.*}.err)"
        tail -20 "$OUT/${log%.*}.err"
        exit 1
    fi
    python3 - "$OUT/$log" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        b = json.loads(ln); c = b["config"]
        print(f"    value {b['value']:.0f} {b['unit']}, {b['ms_per_step']:.3f} ms per pass, n_gpus {b['n_gpus']}; {c['parallelism']}")
        for d in c.get("degraded") or []:
            print("    fell back:", d)
PY
}
step "1. pytest tests/test_multi_gpu.py" pytest.log python3 -m pytest tests/test_multi_gpu.py -x -q -s
step "2. bench.py --gpus 2" bench_2.json python3 bench.py --gpus 2
step "3. bench.py --gpus $N" bench_$N.json python3 bench.py --gpus "$N"
step "4. bench.py --gpus $N --config 5" bench_${N}_config5.json python3 bench.py --gpus "$N" --config 5
for f in comm resident_strips resident_strips,strips; do
    SRPS_FORCE_FAIL=$f step "5. bench.py --gpus $N --config 5 with SRPS_FORCE_FAIL=$f" "bench_${N}_config5_forced_${f//,/_}.json" python3 bench.py --gpus "$N" --config 5 --steps 2 --no-total-solve
done
echo "FIRST CONTACT: every step passed; lines under $OUT"
