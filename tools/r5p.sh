mkdir -p gpurun_out/r5p; O=gpurun_out/r5p
timeout 400 python tools/stress_resident.py 240 > $O/stress_resident.log 2>&1; echo rc=$? >> $O/stress_resident.log
timeout 250 python tools/stress_group.py 120 > $O/stress_group.log 2>&1; echo rc=$? >> $O/stress_group.log
bash tools/fault_sequence.sh > $O/fault_sequence.log 2>&1
tail -3 $O/stress_resident.log $O/stress_group.log; cat $O/fault_sequence.log; tail -3 gpurun_out/fh_D.log gpurun_out/fh_E_tail.log
