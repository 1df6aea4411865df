mkdir -p gpurun_out/r5d; O=gpurun_out/r5d
python -m pytest tests -x -q -m gpu --durations=15 > $O/pytest.log 2>&1; echo rc=$? >> $O/pytest.log
bash tools/ab_pass.sh > $O/ab_light.log 2>&1
python tools/setup_time.py > $O/setup_default.log 2>&1
SRPS_XFER_WC=1 python tools/setup_time.py > $O/setup_wc.log 2>&1
SRPS_XFER_THREADS=4 python tools/setup_time.py > $O/setup_t4.log 2>&1
SRPS_XFER_THREADS=12 python tools/setup_time.py > $O/setup_t12.log 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
grep -v "Gloo\|amdgpu\|socket.cpp" $O/pytest.log | tail -25; cat $O/ab_light.log | cut -c1-260
for f in setup_default setup_wc setup_t4 setup_t12; do echo == $f; grep -E "floats set-up [123]|bytes set-up [123]" $O/$f.log; done
