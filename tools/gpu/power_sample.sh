(CLAIR_AMD_PROJ2_GROUPS=${G:-4} timeout 120 python bench.py --steps 60000 --streams 3 > /tmp/b.json 2>/dev/null) &
BP=$!
sleep 14
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" ; sleep 0.5; done
wait $BP
python -c "
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
