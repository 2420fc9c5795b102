#!/bin/sh
# the two-ranks-on-one-GPU plumbing test; call_var from binary records end to end on the final build
cd "$(dirname "$0")/../.."
O=gpurun_out
timeout 900 python -m pytest tests/test_comm_gpu.py -m gpu -q -x > $O/r05_s6_tests.txt 2>&1; echo "pytest rc $?" >> $O/r05_s6_tests.txt
timeout 900 python tools/e2e_binary_sweep.py 2000000 4096 1024 > $O/r05_e2e_binary.txt 2>&1
tail -25 $O/r05_s6_tests.txt; tail -14 $O/r05_e2e_binary.txt
