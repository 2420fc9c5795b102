#!/bin/sh
# re-baseline after the arithmetic change: weight / input parity sweep, run-to-run and mode-to-mode bit identity (final build)
cd "$(dirname "$0")/../.."
O=gpurun_out
timeout 900 python tools/parity_sweep.py > $O/r05_parity_sweep.txt 2>&1
timeout 600 python tools/gpu/determinism_stress.py 20 > $O/r05_determinism.txt 2>&1
timeout 900 python tools/gpu/cross_mode_stress.py 2 >> $O/r05_determinism.txt 2>&1
cat $O/r05_parity_sweep.txt | cut -c1-200; tail -12 $O/r05_determinism.txt
