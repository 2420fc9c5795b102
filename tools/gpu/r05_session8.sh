#!/bin/sh
# two probes behind LABNOTES A2's arguments, same box, alternating, sustained leg:
#   lstm2_plus700 = 704 idle cycles (44 x s_nop 15) appended to every step of lstm32_kernel<false>: +13 % of its CU-time, not one joule of work more
#   lstm1_noxlo   = LSTM1's eight w_hi.x_lo MFMAs per step replaced by an s_nop of their issue cost (x_lo == 0 for the synthetic counts: same results)
cd "$(dirname "$0")/../.."
O=gpurun_out
timeout 1200 tools/gpu/ab_multi.sh -r 3 final=- lstm2_plus700=build_ab/libclair_amd_lstm2_plus700.so lstm1_noxlo=build_ab/libclair_amd_lstm1_noxlo.so > $O/r05_ab_probes.txt 2>&1
cat $O/r05_ab_probes.txt
