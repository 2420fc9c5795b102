#!/bin/sh
# the -m gpu suite with per-test durations (round 5 saw 850 s where round 4 took 185 s: which tests?)
cd "$(dirname "$0")/../.."
O=gpurun_out
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=60 > $O/r05_s4_gputests.txt 2>&1; echo "pytest rc $? in $(( $(date +%s) - S )) s" >> $O/r05_s4_gputests.txt
grep -A65 "slowest" $O/r05_s4_gputests.txt | head -80; tail -4 $O/r05_s4_gputests.txt
