run() { timeout 200 python bench.py --streams $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2 streams $1', d['value'])"; }
for g in 1 2; do for st in 3 5 6 7 9; do CLAIR_AMD_PROJ2_GROUPS=$g run $st "groups=$g"; done; done
