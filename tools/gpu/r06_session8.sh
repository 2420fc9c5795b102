#!/bin/sh
# round 6, GPU call 8: what a per-step exchange of h between the CUs of one XCD costs (tools/ubench/xcu_exchange.hip) -- the number DESIGN.md section 8's
# two-/four-CU LSTM2 stands or falls with.
cd "$(dirname "$0")/../.."
O=gpurun_out
mkdir -p $O
X=tools/ubench/xcu_exchange
[ -x $X ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $X.hip -o $X
{
echo "# tools/ubench/xcu_exchange <P> <groups per XCD> <steps> <mfma per step> <release>; 48 MFMAs = the x-part of one step per wave in the four-CU split, 96 in the two-CU split"
for rel in 0 1; do
  echo "## four CUs per group, one group per XCD (32 CUs busy)"; $X 4 1 3300 48 $rel
  echo "## four CUs per group, eight groups per XCD (the whole chip)"; $X 4 8 3300 48 $rel
  echo "## two CUs per group, sixteen groups per XCD (the whole chip)"; $X 2 16 3300 96 $rel
done
echo "## four CUs per group, whole chip, 24 MFMAs (half of the x-part left to hide behind)"; $X 4 8 3300 24 0
echo "## four CUs per group, whole chip, 33 steps (one forward pass: start-up included)"; $X 4 8 33 48 0
} > $O/r06_xcu_exchange.txt 2>&1
cat $O/r06_xcu_exchange.txt
