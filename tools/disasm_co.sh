#!/bin/bash
# usage: extract_co.sh lib.so out.s  -- disassemble the gfx950 code object embedded in a HIP shared library
set -e
lib=$1; out=$2
tmp=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin $lib $tmp/fatbin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/co.o
/opt/rocm/lib/llvm/bin/llvm-objdump -d $tmp/co.o > $out
rm -rf $tmp
