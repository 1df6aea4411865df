mkdir -p gpurun_out/r5i; O=gpurun_out/r5i
python tools/setup_outliers.py > $O/outliers_default.log 2>&1
python tools/setup_outliers.py SRPS_XFER_THREADS=4 > $O/outliers_t4.log 2>&1
python bench.py --no-legs --no-cpu-baseline > $O/bench.json 2> $O/bench.err
grep "set-up" $O/outliers_default.log; echo; grep "set-up" $O/outliers_t4.log
