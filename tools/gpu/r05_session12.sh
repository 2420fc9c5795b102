#!/bin/sh
# projection workgroup groups per XCD under four lanes (2 = 64 workgroups, 3 = 96 (default), 4 = 128), same box, alternating
cd "$(dirname "$0")/../.."
O=gpurun_out
timeout 1200 tools/gpu/ab_multi.sh -r 2 g3=- g2=-,CLAIR_AMD_PROJ2_GROUPS=2 g4=-,CLAIR_AMD_PROJ2_GROUPS=4 g5=-,CLAIR_AMD_PROJ2_GROUPS=5 > $O/r05_ab_proj2_groups.txt 2>&1
cut -c1-120 $O/r05_ab_proj2_groups.txt
