#!/bin/sh
# One box's sample for the round-2 outlier hunt (profiles/r03_excursion_hunt.txt): the run-twice test of the tool's configuration, the
# instrumented 3 x 200 000 concordance twice (once keeping the LSTM taps of every batch), the production build against the wait-all build.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
TAG=${1:-x}
python -m pytest tests/test_parity_gpu.py -m gpu -q -k "twice_in_one_process" 2>&1 | tail -1
python tools/gt_concordance.py --json gpurun_out/exc_${TAG}_a.json > gpurun_out/exc_${TAG}_a.log 2>&1; tail -1 gpurun_out/exc_${TAG}_a.log
python tools/gt_concordance.py --taps --json gpurun_out/exc_${TAG}_b.json > gpurun_out/exc_${TAG}_b.log 2>&1; tail -1 gpurun_out/exc_${TAG}_b.log
python tools/gpu/waitall_compare.py 6 1 2>&1 | tail -1
grep -h '"max_abs_dp"' gpurun_out/exc_${TAG}_a.log gpurun_out/exc_${TAG}_b.log | cut -c1-135
