timeout 1300 tools/gpu/contended_sample.sh final2 3 100000 2 > /dev/null 2>&1
cat gpurun_out/r04_contended_final2.txt
for i in 1 2 3 4 5 6; do
python bench.py --steps 196 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('gpu_state',{}).get('value_full_config',{})
print('run $i: resident', d['value'], d['value_full_config'], 'float32', d['value_boundary'], d['value_boundary_full_config'], 'int16', d['value_boundary_int16'], d.get('value_boundary_int16_full_config'), 'sclk/power of the whole-set leg', g.get('sclk_mhz'), g.get('power_w'))"
done | tee gpurun_out/r04_bench_repeats_final2.txt
