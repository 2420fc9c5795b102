#!/bin/sh
# Same-box A/B of any number of variants, alternating:  ab_multi.sh [-r rounds] [-a "bench args"] name=lib[,ENV=VAL...] ...
# A variant is a build of the engine (path relative to the repository root, or "-" for the tree's own) plus environment
# variables; every round runs every variant once (bench.py: the contract's K steps, then the >= 2 s sustained leg);
# one line per run: name, value (burst), value_sustained, ms/step, board power and clock during the sustained leg.
cd "$(dirname "$0")/../.."
ROUNDS=3
ARGS="--steps 200 --warmup 8 --sustained-seconds 2"
BOUNDARY="--boundary-slots 0"          # -b: keep the host-array boundary legs (value_boundary, value_boundary_int16)
while getopts r:a:b o; do case $o in r) ROUNDS=$OPTARG;; a) ARGS=$OPTARG;; b) BOUNDARY="";; esac; done
shift $((OPTIND - 1))
i=1
while [ $i -le $ROUNDS ]; do
  for v in "$@"; do
    name=${v%%=*}; rest=${v#*=}
    lib=${rest%%,*}; envs=""
    case $rest in *,*) envs=$(echo "${rest#*,}" | tr ',' ' ');; esac
    [ "$lib" = "-" ] && libenv="" || libenv="CLAIR_AMD_LIB=$PWD/$lib"
    out=$(env $libenv $envs timeout 300 python bench.py $ARGS --no-cpu-baseline --gt-candidates 0 --full-candidates 0 $BOUNDARY 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=d['gpu_state'].get('value_sustained') or d['gpu_state'].get('value') or {}
b=d.get('boundary') or {}
print('value %.0f sustained %s ms/step %s power %s W sclk %s MHz parity %.2e boundary f32 %s i16 %s alone %s' % (d['value'], d.get('value_sustained'), (d.get('sustained') or {}).get('ms_per_step'), g.get('power_w'), g.get('sclk_mhz'), d['parity_max_abs_err'] if d['parity_max_abs_err'] is not None else float('nan'),
      (b.get('float32') or {}).get('value'), (b.get('int16') or {}).get('value'), {k: v for k, v in d['kernels_alone_ms'].items() if v}))")
    echo "round $i $name: $out"
  done
  i=$((i + 1))
done
