for lb in 0 768 0 768; do
timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-total-solve --option light_blocks=$lb 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('light_blocks=$lb', round(d['value']), round(d['ms_per_step'],3), d['energies'][-1])"
done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/solve_time.py 960 1280 2 20 ragged 2>&1 | tail -1
