#!/bin/sh
# round 6, GPU call 3: the CCS batch-4096 and Illumina batch-8192 profile sets, the cache-hint probe (a1 / a2 stored non-temporally, a2 read with nt),
# the wait-all comparison, determinism, and the end-to-end runs (binary records, BAM region) on the round-6 tree.
cd "$(dirname "$0")/../.."
O=gpurun_out
mkdir -p $O
S=$(date +%s)
tools/gpu/profile_r06.sh ccs_b4096 48 --platform pacbio_ccs --batch 4096 > $O/r06_s3_profile_ccs.log 2>&1; tail -6 $O/r06_s3_profile_ccs.log | cut -c1-300
tools/gpu/profile_r06.sh illumina_b8192 24 --platform illumina --batch 8192 > $O/r06_s3_profile_illumina.log 2>&1; tail -6 $O/r06_s3_profile_illumina.log | cut -c1-300
{
echo "# cache-hint probe (tools/gpu/nt_variants.sh): a1 (LSTM1 -> projection) and a2 (LSTM2 -> l3l4) stored non-temporally like zx; a2read: l3l4's LDS-DMA of a2 with nt"
echo "# same box, alternating, ONT batch 1024; results are identical bits (hints only)"
tools/gpu/ab_multi.sh -r 3 tree=- nt_a1=build_ab/libclair_amd_nt_a1.so nt_a2=build_ab/libclair_amd_nt_a2.so nt_a1a2=build_ab/libclair_amd_nt_a1a2.so nt_a2read=build_ab/libclair_amd_nt_a2read.so
} > $O/r06_ab_cache_hints.txt 2>&1
cat $O/r06_ab_cache_hints.txt
timeout 600 python tools/gpu/waitall_compare.py 6 1 > $O/r06_waitall.txt 2>&1; tail -3 $O/r06_waitall.txt
timeout 400 python tools/gpu/determinism_stress.py 200 > $O/r06_determinism.txt 2>&1; tail -3 $O/r06_determinism.txt
timeout 900 python tools/e2e_binary_sweep.py 2000000 4096 1024 > $O/r06_e2e_binary.txt 2>&1; tail -10 $O/r06_e2e_binary.txt
timeout 600 python tools/e2e_bam_bench.py 10000000 > $O/r06_e2e_bam.txt 2>&1; tail -10 $O/r06_e2e_bam.txt
echo "session wall $(( $(date +%s) - S )) s"
