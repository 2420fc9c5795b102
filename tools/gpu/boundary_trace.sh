#!/bin/sh
# rocprofv3 kernel + memory-copy trace of the boundary loop.  usage: boundary_trace.sh <tag> <slots> [counts]   (engine modes via the environment)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=$1; SLOTS=$2; KIND=$3
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
rm -rf $O/trace_$TAG
( cd $R && timeout 120 python tools/gpu/boundary_trace.py $SLOTS 240 $KIND ) > $O/trace_${TAG}_plain.txt 2>&1
( cd $R && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/trace_$TAG -o t -- python tools/gpu/boundary_trace.py $SLOTS 240 $KIND ) > $O/trace_${TAG}.log 2>&1
DB=$(ls $O/trace_$TAG/*.db 2>/dev/null | head -1)
( cat $O/trace_${TAG}_plain.txt; grep -E "boundary|resident" $O/trace_${TAG}.log | sed 's/^/traced: /'; python $R/tools/boundary_timeline.py $DB --dump ) > $O/timeline_$TAG.txt 2>&1
rm -rf $O/trace_$TAG
