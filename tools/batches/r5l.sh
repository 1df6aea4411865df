mkdir -p gpurun_out/r5l; O=gpurun_out/r5l
for i in 1 2 3; do
python tools/pass_time.py 2048 4 20 10 >> $O/pass.jsonl 2>>$O/pass.err
python tools/pass_time.py 2048 4 20 10 image_tiles=1 >> $O/pass.jsonl 2>>$O/pass.err
done
cut -c1-260 $O/pass.jsonl
