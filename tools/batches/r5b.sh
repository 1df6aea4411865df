mkdir -p gpurun_out/r5b; O=gpurun_out/r5b
python -m pytest tests/test_gpu_strips.py -x -q -s -m gpu -k "one_process or recognised or ipc_mapped or bit_for_bit" > $O/strips.log 2>&1; echo rc=$? >> $O/strips.log
python -m pytest tests/test_gpu_edge_and_scale.py tests/test_gpu_parity.py tests/test_gpu_image_store.py -x -q -m gpu > $O/light.log 2>&1; echo rc=$? >> $O/light.log
for i in 1 2; do
python tools/pass_time.py 2048 4 20 10 light_run=0 >> $O/pass.jsonl 2>>$O/pass.err
python tools/pass_time.py 2048 4 20 10 light_run=1 >> $O/pass.jsonl 2>>$O/pass.err
done
python tools/setup_time.py > $O/setup_default.log 2>&1
SRPS_XFER_NT=0 python tools/setup_time.py > $O/setup_nt0.log 2>&1
SRPS_XFER_THREADS=8 python tools/setup_time.py > $O/setup_t8.log 2>&1
SRPS_XFER_THREADS=32 python tools/setup_time.py > $O/setup_t32.log 2>&1
python tools/setup_time.py 2048 20 pin_uploads=1 > $O/setup_pin.log 2>&1
python -m pytest tests/test_gpu_whole_solve.py -x -q -s -m gpu -k config3 > $O/whole.log 2>&1; echo rc=$? >> $O/whole.log
tail -3 $O/strips.log $O/light.log $O/whole.log; cat $O/pass.jsonl
