#!/bin/sh
# The round-2 excursion hunt UNDER CONTENTION: every earlier sample ran with the GPU to itself.  Per run, on the same GPU
# at the same time:
#   * tools/ubench/hbm_hog            -- a process streaming 2 GiB copies (HBM saturated, two workgroups on every CU);
#   * tools/gpu/fused_corunner.py x K -- K processes on the fused one-slot layer-2 path, each bit-comparing every pass with its first and, at
#                                        the end, with the two-launch path (queue oversubscribed: recover_fused has to cope in the wild);
#   * the sample itself               -- tools/gt_concordance.py (3 x N candidates against the oracle, dissect-on-excursion) and
#                                        tools/gpu/waitall_compare.py (production build vs the wait-all CHECK build, bit for bit).
# usage: contended_sample.sh <tag> [runs=10] [n=100000] [fused co-runners=2]      -> gpurun_out/r04_contended_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
TAG=${1:-x}; RUNS=${2:-10}; N=${3:-100000}; K=${4:-2}
O=gpurun_out/r04_contended_$TAG.txt
mkdir -p gpurun_out
[ -x tools/ubench/hbm_hog ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/hbm_hog.hip -o tools/ubench/hbm_hog
{
echo "# contended sample $TAG: $RUNS runs, 3 x $N candidates each, co-runners: hbm_hog + $K x fused_corunner; $(date -u +%Y-%m-%dT%H:%MZ)"
for r in $(seq 1 $RUNS); do
  tools/ubench/hbm_hog 600 > /tmp/hog_$r.log 2>&1 &
  HOG=$!
  PIDS=""
  for k in $(seq 1 $K); do
    timeout 900 python tools/gpu/fused_corunner.py 45 "run $r co-runner $k" > /tmp/co_${r}_$k.log 2>&1 &
    PIDS="$PIDS $!"
  done
  sleep 3
  timeout 600 python tools/gt_concordance.py --n $N --json gpurun_out/contended_${TAG}_$r.json > /tmp/gt_$r.log 2>&1
  GT=$?
  timeout 600 python tools/gpu/waitall_compare.py 1 1 > /tmp/wa_$r.log 2>&1
  WA=$?
  for p in $PIDS; do wait $p; done
  kill $HOG 2>/dev/null; wait $HOG 2>/dev/null
  echo "run $r: gt_concordance rc $GT: $(grep -h 'max_abs_dp' /tmp/gt_$r.log | grep -o 'max_abs_dp\": [0-9.e-]*' | tr '\n' ' ') $(tail -1 /tmp/gt_$r.log | cut -c1-160)"
  echo "run $r: waitall_compare rc $WA: $(grep RESULT /tmp/wa_$r.log | cut -c1-220)"
  for k in $(seq 1 $K); do echo "run $r: $(tail -1 /tmp/co_${r}_$k.log | cut -c1-220)"; done
  grep -h "clair_amd: the fused" /tmp/co_${r}_*.log | head -2
done
} > $O 2>&1
cat $O
