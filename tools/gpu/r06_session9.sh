#!/bin/sh
# round 6, GPU call 9: cache-policy bits of the projection's zx stores (today: nt): sc1 nt | sc0 sc1 nt | sc0 nt | sc1 | sc0 sc1 -- same box, alternating, identical bits.
cd "$(dirname "$0")/../.."
O=gpurun_out
mkdir -p $O
{
echo "# zx stores of gemm_split_kernel by cache-policy bits (inline asm global_store_dwordx4 ... <bits>; the tree uses the nontemporal builtin = nt)"
tools/gpu/ab_multi.sh -r 2 tree=- sc1_nt=build_ab/libclair_amd_zxst_sc1_nt.so sc0_sc1_nt=build_ab/libclair_amd_zxst_sc0_sc1_nt.so sc0_nt=build_ab/libclair_amd_zxst_sc0_nt.so sc1=build_ab/libclair_amd_zxst_sc1.so sc0_sc1=build_ab/libclair_amd_zxst_sc0_sc1.so
} > $O/r06_ab_zx_store_bits.txt 2>&1
sed -e 's/ parity \([^ ]*\) boundary f32 None i16 None/ parity \1/' $O/r06_ab_zx_store_bits.txt
