#!/bin/sh
# Two-tile LSTM2 at batch 1024 against the number of slots: the two-tile kernel spends 20 % fewer CU-cycles per tile-step but is a
# longer launch (32 workgroups x 128 us); with more chains in flight its latency may stop mattering.
cd "$(dirname "$0")/../.."
for pair in 0 1; do
  for streams in 3 4 5 6; do
    v=$(CLAIR_AMD_LSTM2_PAIR=$pair timeout 200 python bench.py --steps 1200 --warmup 8 --streams $streams --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k: round(v['ms_mean'], 4) for k, v in d['kernels_in_flight_ms'].items() if (v['ms_mean'] or 0) > 0.001})")
    echo "pair=$pair streams=$streams: $v"
  done
done
