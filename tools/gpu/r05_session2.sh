#!/bin/sh
# Round 5, second GPU call: the whole GPU suite on the tree; l3l4 with the a2 tiles let in by the consumers against the build before
# (same box, alternating) and its phase clocks; staging threads of the host-array boundary.
cd "$(dirname "$0")/../.."
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/r05_s2_gputests.txt 2>&1; echo "pytest rc $?" >> $O/r05_s2_gputests.txt
timeout 900 tools/gpu/ab_multi.sh -r 3 v2=build_ab/libclair_amd_v2.so l34dma=- > $O/r05_ab_l34dma.txt 2>&1
timeout 300 python tools/gpu/l34_stamps.py 1024 > $O/r05_l34_stamps.txt 2>&1
timeout 600 tools/gpu/ab_multi.sh -r 2 -b -a "--steps 600 --warmup 8 --sustained-seconds 0" st2=- st3=-,CLAIR_AMD_STAGING_THREADS=3 st4=-,CLAIR_AMD_STAGING_THREADS=4 > $O/r05_ab_staging.txt 2>&1
tail -15 $O/r05_s2_gputests.txt; cat $O/r05_ab_l34dma.txt; tail -8 $O/r05_l34_stamps.txt; cat $O/r05_ab_staging.txt
