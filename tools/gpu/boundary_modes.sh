#!/bin/sh
# The submit/wait loop of tools/gpu/boundary_probe.py under the engine's copy-stream modes, hardware-queue counts and copy engines.
cd "$(dirname "$0")/../.."
for hwq in 4 8; do for sdma in 1 0; do for mode in slot two lane; do for stage in 1 0; do
  echo "### GPU_MAX_HW_QUEUES=$hwq HSA_ENABLE_SDMA=$sdma CLAIR_AMD_COPY_STREAMS=$mode CLAIR_AMD_ASYNC_STAGING=$stage"
  GPU_MAX_HW_QUEUES=$hwq HSA_ENABLE_SDMA=$sdma CLAIR_AMD_COPY_STREAMS=$mode CLAIR_AMD_ASYNC_STAGING=$stage timeout 120 python tools/gpu/boundary_probe.py --no-raw --slots ${SLOTS:-3,6} --modes ${MODES:-pinned,int16} 2>&1 | grep slots
done; done; done; done
