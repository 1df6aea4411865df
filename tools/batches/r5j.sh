mkdir -p gpurun_out/r5j; O=gpurun_out/r5j
VARIANTS=srmeetsps-cuda_amd/variants_l bash tools/ab_pass.sh > $O/ab_light.log 2>&1
cut -c1-230 $O/ab_light.log
