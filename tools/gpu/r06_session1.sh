#!/bin/sh
# round 6, GPU call 1: mint the GT tie set (tests/golden/gt_ties.npz), the two-tile LSTM2 kernel without scratch (bit-identity test + A/B
# against the round-5 build at the two configurations it is the default for), fabric sensitivity, the nozx probe on the current kernels.
cd "$(dirname "$0")/../.."
O=gpurun_out
mkdir -p $O
S=$(date +%s)
timeout 900 python tools/gt_concordance.py --n 200000 --tightest 8 --ties $O/gt_ties.npz --json $O/r06_gt_concordance.json > $O/r06_gt_concordance.txt 2>&1
echo "gt_concordance rc $? in $(( $(date +%s) - S )) s" >> $O/r06_gt_concordance.txt
grep -E "FLIP|^\{|rc " $O/r06_gt_concordance.txt | cut -c1-600
timeout 600 python -m pytest tests/test_parity_gpu.py -q -x -k "two_tile or pair or gt_concordance_65k" > $O/r06_s1_pytests.txt 2>&1; tail -5 $O/r06_s1_pytests.txt
{
echo "# two-tile LSTM2 kernel: round-5 build (16 B of scratch per lane) against the tree (none); same box, alternating -- tools/gpu/r06_session1.sh"
echo "## CCS profile, batch 4096"
tools/gpu/ab_multi.sh -r 3 -a "--platform pacbio_ccs --batch 4096 --steps 48 --warmup 4 --sustained-seconds 2" r05=build_ab/libclair_amd_r05.so r06=-
echo "## Illumina profile, batch 8192"
tools/gpu/ab_multi.sh -r 3 -a "--platform illumina --batch 8192 --steps 24 --warmup 4 --sustained-seconds 2" r05=build_ab/libclair_amd_r05.so r06=-
} > $O/r06_ab_pair_noscratch.txt 2>&1
cat $O/r06_ab_pair_noscratch.txt
tools/gpu/fabric_sensitivity.sh $O/r06_fabric_sensitivity.txt
{
echo "# nozx probe on the round-6 kernels (tools/gpu/nozx_variants.sh): zx addressed through a ring of F x 64 KB per direction; nt0 = plain stores / loads"
echo "# (results are garbage, timing only; zxfold100000 = the production addressing)"
tools/gpu/ab_multi.sh -r 2 -a "--steps 200 --warmup 8 --sustained-seconds 2 --no-parity" tree=- prod_nt1=build_ab/libclair_amd_zxfold100000_nt1.so prod_nt0=build_ab/libclair_amd_zxfold100000_nt0.so \
   ring16_nt1=build_ab/libclair_amd_zxfold16_nt1.so ring16_nt0=build_ab/libclair_amd_zxfold16_nt0.so ring1_nt0=build_ab/libclair_amd_zxfold1_nt0.so
} > $O/r06_nozx_probe.txt 2>&1
cat $O/r06_nozx_probe.txt
echo "session wall $(( $(date +%s) - S )) s"
