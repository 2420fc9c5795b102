tools/gpu/profile_r04.sh ont_b1024 196 > gpurun_out/r04_profile_a.log 2>&1
tools/gpu/profile_r04.sh ccs_b4096 48 --batch 4096 --platform pacbio_ccs --unique-batches 4 > gpurun_out/r04_profile_b.log 2>&1
tools/gpu/profile_r04.sh illumina_b8192 24 --batch 8192 --platform illumina --unique-batches 4 > gpurun_out/r04_profile_c.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_driver_like_bench.json 2> gpurun_out/r04_driver_like_bench.err
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
for f in ont_b1024 ccs_b4096 illumina_b8192; do head -c 300 gpurun_out/r04_${f}_bench.json; echo; done; head -c 300 gpurun_out/r04_driver_like_bench.json
