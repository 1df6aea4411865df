mkdir -p gpurun_out/r5o; O=gpurun_out/r5o
VARIANTS=srmeetsps-cuda_amd/variants_l bash tools/ab_pass.sh > $O/ab_albedo.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r5o/ab_albedo.log'):
    n, j = l.split(': ', 1)
    try:
        d = json.loads(j)
        print(n, d['ms_per_pass'], d['phase_ms']['albedo_sweep'], d['phase_ms']['energy'])
    except Exception as e:
        print(l[:150])
PY
