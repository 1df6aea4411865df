mkdir -p gpurun_out/r5c; O=gpurun_out/r5c
timeout 300 python tools/oracle_threads_probe.py > $O/probe_bare.log 2>&1
timeout 300 python tools/oracle_threads_probe.py torch > $O/probe_torch.log 2>&1
timeout 300 python tools/oracle_threads_probe.py torch cuda > $O/probe_cuda.log 2>&1
OMP_WAIT_POLICY=PASSIVE timeout 300 python tools/oracle_threads_probe.py torch cuda > $O/probe_cuda_passive.log 2>&1
python tools/setup_time.py > $O/setup_default.log 2>&1
SRPS_XFER_STREAMS=2 python tools/setup_time.py > $O/setup_s2.log 2>&1
SRPS_XFER_CHUNK_MB=32 python tools/setup_time.py > $O/setup_c32.log 2>&1
SRPS_XFER_CHUNK_MB=32 SRPS_XFER_STREAMS=2 python tools/setup_time.py > $O/setup_c32s2.log 2>&1
SRPS_XFER_CHUNK_MB=64 python tools/setup_time.py > $O/setup_c64.log 2>&1
SRPS_XFER_CHUNK_MB=8 SRPS_XFER_STREAMS=2 python tools/setup_time.py > $O/setup_c8s2.log 2>&1
bash tools/ab_variants.sh "2048 4 full" > $O/ab.log 2>&1
python -m pytest tests/test_gpu_strips.py -x -q -m gpu -k "recognised" > $O/strips.log 2>&1; echo rc=$? >> $O/strips.log
cat $O/probe_bare.log; tail -8 $O/probe_cuda.log; cat $O/ab.log
