for i in 1 2; do
for L in abtmp/lib_old.so clair_amd/libclair_amd.so; do
CLAIR_AMD_LIB=$PWD/$L timeout 200 python bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L', d['value'], d['kernels_single_stream_ms'])"
done; done
