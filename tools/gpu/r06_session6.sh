#!/bin/sh
# round 6, GPU call 6: evidence for DESIGN.md section 8 -- (a) L2 <-> fabric traffic of the fused layer-2 launch with non-temporal and with plain zx (does the
# read half of the round trip vanish when the stores are plain?); (b) the fused launch by lanes in flight x projection groups, plain zx: is there a lane
# structure in which it pays?  (c) the driver's command line twice more (box-to-box / run-to-run spread of the three resident rates).
cd "$(dirname "$0")/../.."
O=gpurun_out
mkdir -p $O
{
echo "# fused layer-2 launch, one slot, PMC FETCH_SIZE x2 + WRITE_SIZE per launch (tools/gpu/fused_pmc.sh)"
echo "## zx stored non-temporally (the tree)"
STREAMS=1 tools/gpu/fused_pmc.sh --gt-candidates 0 --sustained-seconds 0 --boundary-slots 0 --full-candidates 0 2>&1 | tail -9
echo "## zx with plain stores / loads (build_ab/libclair_amd_zxfold100000_nt0.so)"
CLAIR_AMD_LIB=$PWD/build_ab/libclair_amd_zxfold100000_nt0.so STREAMS=1 tools/gpu/fused_pmc.sh --gt-candidates 0 --sustained-seconds 0 --boundary-slots 0 --full-candidates 0 --no-parity 2>&1 | tail -9
} > $O/r06_fused_traffic.txt 2>&1
cat $O/r06_fused_traffic.txt
{
echo "# the fused launch with plain zx, by lanes in flight (--streams) and projection groups per XCD (CLAIR_AMD_FUSED_GROUPS; producers = 32 x groups, consumers 64 at batch 1024)"
for st in 2 3 4; do
  echo "## $st lanes"
  tools/gpu/ab_multi.sh -r 1 -a "--steps 200 --warmup 8 --sustained-seconds 2 --streams $st" two_launches=- \
     fused_g2=build_ab/libclair_amd_zxfold100000_nt0.so,CLAIR_AMD_LSTM2_FUSED=1,CLAIR_AMD_FUSED_GROUPS=2 \
     fused_g3=build_ab/libclair_amd_zxfold100000_nt0.so,CLAIR_AMD_LSTM2_FUSED=1,CLAIR_AMD_FUSED_GROUPS=3 \
     fused_g4=build_ab/libclair_amd_zxfold100000_nt0.so,CLAIR_AMD_LSTM2_FUSED=1,CLAIR_AMD_FUSED_GROUPS=4
done
} > $O/r06_fused_lanes.txt 2>&1
sed -e 's/ parity [^ ]* boundary f32 None i16 None//' $O/r06_fused_lanes.txt
for k in 2 3; do
  S=$(date +%s)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_driver_like_bench_$k.json 2> /dev/null
  python -c "
import json; d=json.loads(open('$O/r06_driver_like_bench_$k.json').read().strip().splitlines()[-1]); r=d['roofline']
print('driver-like run $k:', d['value'], d['value_full_config'], d['value_sustained'], 'wall', d['bench_wall_s'], 'frac', r['frac'], r['frac_rocprof'], 'flips', [p['gt_flips'] for p in d['gt_concordance_200k']['platforms'].values()], 'not excused', [p['flips_not_excused'] for p in d['gt_concordance_200k']['platforms'].values()])"
done
