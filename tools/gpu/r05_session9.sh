#!/bin/sh
# experiment: LSTM1 of the next batch launched with hipExtAnyOrderLaunch (may start while the lane's previous tail is still running)
cd "$(dirname "$0")/../.."
O=gpurun_out
timeout 1200 tools/gpu/ab_multi.sh -r 3 final=- anyorder_off=build_ab/libclair_amd_anyorder.so anyorder_on=build_ab/libclair_amd_anyorder.so,CLAIR_AMD_ANYORDER=1 > $O/r05_ab_anyorder.txt 2>&1
cat $O/r05_ab_anyorder.txt
