#!/bin/sh
# second sweep: one incoming copy stream, results leaving on the lane's stream (copy engine or kernel), by slots, lanes and hardware queues
cd "$(dirname "$0")/../.."
run() { echo "### $*"; env "$@" timeout 120 python tools/gpu/boundary_probe.py --no-raw --slots ${SLOTS:-3,6,9} --modes ${MODES:-pageable,pinned,int16} 2>&1 | grep slots; }
run CLAIR_AMD_COPY_STREAMS=in CLAIR_AMD_D2H=sdma
run CLAIR_AMD_COPY_STREAMS=in CLAIR_AMD_D2H=kernel
run CLAIR_AMD_COPY_STREAMS=in CLAIR_AMD_D2H=kernel CLAIR_AMD_ASYNC_STAGING=0
run CLAIR_AMD_COPY_STREAMS=in CLAIR_AMD_D2H=kernel GPU_MAX_HW_QUEUES=8
run CLAIR_AMD_COPY_STREAMS=in CLAIR_AMD_D2H=kernel GPU_MAX_HW_QUEUES=8 CLAIR_AMD_LANES=4
run CLAIR_AMD_COPY_STREAMS=in CLAIR_AMD_D2H=sdma GPU_MAX_HW_QUEUES=8 CLAIR_AMD_LANES=4
run CLAIR_AMD_COPY_STREAMS=two CLAIR_AMD_D2H=kernel
