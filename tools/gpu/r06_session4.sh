#!/bin/sh
# round 6, GPU call 4: cache hints round 2 (a2 handled non-temporally on both sides, + the input), and the fused layer-2 launch revisited with four
# lanes: as it is (non-temporal zx stores: the block leaves the L2 at once) and with PLAIN zx stores / loads (a write-back L2 keeps the block for the
# reader on the same XCD: does the read half of the round trip vanish?).
cd "$(dirname "$0")/../.."
O=gpurun_out
mkdir -p $O
{
echo "# cache hints, round 2 (tools/gpu/nt_variants.sh; identical bits): a2r = l3l4's LDS-DMA of a2 with nt, a2w = LSTM2's a2 stores non-temporal, xr = LSTM1's input loads non-temporal"
tools/gpu/ab_multi.sh -r 3 tree=- a2r=build_ab/libclair_amd_nt_a2r.so a2w+a2r=build_ab/libclair_amd_nt_a2wa2r.so a2w+a2r+xr=build_ab/libclair_amd_nt_a2wa2rxr.so
echo "## CCS batch 4096"
tools/gpu/ab_multi.sh -r 2 -a "--platform pacbio_ccs --batch 4096 --steps 48 --warmup 4 --sustained-seconds 2" tree=- a2w+a2r=build_ab/libclair_amd_nt_a2wa2r.so a2w+a2r+xr=build_ab/libclair_amd_nt_a2wa2rxr.so
} > $O/r06_ab_cache_hints2.txt 2>&1
cat $O/r06_ab_cache_hints2.txt
{
echo "# the fused layer-2 launch (opt-in, CLAIR_AMD_LSTM2_FUSED=1) under four lanes: two launches (tree) | fused, zx stored non-temporally (as built) | fused, zx with"
echo "# plain stores and loads (build_ab/libclair_amd_zxfold100000_nt0.so: production addressing, no hints)"
tools/gpu/ab_multi.sh -r 2 two_launches=- fused_nt=-,CLAIR_AMD_LSTM2_FUSED=1 fused_plain=build_ab/libclair_amd_zxfold100000_nt0.so,CLAIR_AMD_LSTM2_FUSED=1 two_launches_plain=build_ab/libclair_amd_zxfold100000_nt0.so
echo "## two slots"
tools/gpu/ab_multi.sh -r 2 -a "--steps 200 --warmup 8 --sustained-seconds 2 --streams 2" two_launches=- fused_nt=-,CLAIR_AMD_LSTM2_FUSED=1 fused_plain=build_ab/libclair_amd_zxfold100000_nt0.so,CLAIR_AMD_LSTM2_FUSED=1
} > $O/r06_ab_fused_plain.txt 2>&1
cat $O/r06_ab_fused_plain.txt
