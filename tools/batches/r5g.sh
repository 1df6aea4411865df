mkdir -p gpurun_out/r5g; O=gpurun_out/r5g
for i in 1 2 3; do
python tools/pass_time.py 2048 4 20 10 light_run=0 >> $O/pass.jsonl 2>>$O/pass.err
python tools/pass_time.py 2048 4 20 10 light_run=1 >> $O/pass.jsonl 2>>$O/pass.err
done
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo rc=$? >> $O/pytest.log
python bench.py --no-legs > $O/bench_nolegs.json 2> $O/bench.err
cut -c1-230 $O/pass.jsonl; grep -v "Gloo\|amdgpu\|socket.cpp" $O/pytest.log | tail -8
