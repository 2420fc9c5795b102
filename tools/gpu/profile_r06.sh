#!/bin/sh
# Round-6 profile of one benchmark configuration.  usage: profile_r06.sh <tag> <steps> <bench.py arguments ...>
#   -> gpurun_out/r06_<tag>_{bench.json, kernel_stats.txt, pmc_hbm_traffic.txt, pmc_mfma_util.txt}   (copied into profiles/ afterwards)
# As profile_r05.sh (+ the traced run's per-kernel means of the timed region go into pmc_traffic.json: roofline.kernel_ms_rocprof); as profile_r04.sh; the traced runs keep the sustained leg short and skip the 3 x 200 000 concordance, the bench line at the END is the full one.
# Counter passes are separate rocprofv3 runs with --kernel-trace only (never with the hip/hsa trace domains).
TAG=$1; STEPS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
[ -f $O/pmc_traffic.json ] || cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json   # merged per batch size, copied back into profiles/ afterwards
BATCH=1024; for a in "$@"; do [ "$prev" = "--batch" ] && BATCH=$a; prev=$a; done
SHORT="--no-cpu-baseline --gt-candidates 0 --sustained-seconds 0.25"
rm -rf $O/prof_r06_${TAG}
timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_r06_${TAG} -o bench -- python $R/bench.py --steps $STEPS $SHORT "$@" > $O/prof_r06_${TAG}.json 2> $O/prof_r06_${TAG}.log
python $R/tools/rocpd_summary.py $O/prof_r06_${TAG}/bench_results.db --phases-from $O/prof_r06_${TAG}.json --resource-usage $R/profiles/r06_kernel_resource_usage.txt --timed-json $O/r06_${TAG}_timed_ms.json > $O/r06_${TAG}_kernel_stats.txt 2>&1
PMCARGS="--steps 8 --warmup 2 --no-cpu-baseline --gt-candidates 0 --sustained-seconds 0 --boundary-slots 0 --full-candidates 0"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_r06_${TAG}_$c
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_r06_${TAG}_$c -o bench -- env BENCH_WARM_STEPS=0 python $R/bench.py $PMCARGS "$@" > $O/pmc_r06_${TAG}_$c.log 2>&1
done
python $R/tools/pmc_summary.py traffic $O/pmc_r06_${TAG}_FETCH_SIZE/bench_results.db $O/pmc_r06_${TAG}_WRITE_SIZE/bench_results.db --batch $BATCH --json $O/pmc_traffic.json --source profiles/r06_${TAG}_pmc_hbm_traffic.txt > $O/r06_${TAG}_pmc_hbm_traffic.txt 2>&1
# the rocprofv3 mean of every kernel in the contract's timed region, next to the traffic entry of the same sources (bench.py: roofline.kernel_ms_rocprof)
python - $O/pmc_traffic.json $BATCH $O/r06_${TAG}_timed_ms.json profiles/r06_${TAG}_kernel_stats.txt <<'PY'
import json, sys
doc = json.load(open(sys.argv[1]))
e = doc["entries"].get(sys.argv[2])
if e is not None:
    e["kernel_ms_rocprof"] = json.load(open(sys.argv[3]))
    e["kernel_ms_rocprof_source"] = sys.argv[4]
    json.dump(doc, open(sys.argv[1], "w"), indent=1)
PY
rm -rf $O/pmc_r06_${TAG}_sq
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_r06_${TAG}_sq -o bench -- env BENCH_WARM_STEPS=0 python $R/bench.py $PMCARGS --streams 1 "$@" > $O/pmc_r06_${TAG}_sq.log 2>&1
python $R/tools/pmc_summary.py mfma $O/pmc_r06_${TAG}_sq/bench_results.db --batch $BATCH --groups 8 > $O/r06_${TAG}_pmc_mfma_util.txt 2>&1
if [ "$TAG" = "ont_b1024" ]; then   # requests "destined for DRAM" against all requests at the L2's fabric side (LABNOTES A2: they do not see the Infinity Cache either)
  for pair in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $pair | cut -d' ' -f1)
    rm -rf $O/pmc_r06_ea_$n
    timeout 400 rocprofv3 --kernel-trace --pmc $pair -d $O/pmc_r06_ea_$n -o bench -- env BENCH_WARM_STEPS=0 python $R/bench.py $PMCARGS "$@" > $O/pmc_r06_ea_$n.log 2>&1
    python - $O/pmc_r06_ea_$n/bench_results.db $pair >> $O/r06_ont_b1024_pmc_ea_requests.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for c in sys.argv[2:]:
    rows = db.execute("select kernel_name, value from counters_collection where counter_name = ? order by start", (c,)).fetchall()
    per = {}
    for name, v in rows:
        per.setdefault(name, []).append(float(v))
    for name, v in sorted(per.items()):
        v = v[2:] if len(v) > 2 else v
        print("%-28s %-60s mean per launch %14.0f  (%d launches)" % (c, name[:60], sum(v) / len(v), len(v)))
PY
    rm -rf $O/pmc_r06_ea_$n
  done
fi
rm -rf $O/prof_r06_${TAG} $O/pmc_r06_${TAG}_FETCH_SIZE $O/pmc_r06_${TAG}_WRITE_SIZE $O/pmc_r06_${TAG}_sq     # the tables above are what is kept (gpurun_out/ comes back only below 64 MiB)
# the bench line LAST: its `traffic` fields are reported only from a table measured on exactly these kernel sources, i.e. the passes above
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
FULL=""; [ "$TAG" = "ont_b1024" ] || FULL="--gt-candidates 0"
timeout 600 python $R/bench.py --steps $STEPS $FULL "$@" > $O/r06_${TAG}_bench.json 2> $O/r06_${TAG}_bench.err
cd $R
head -c 400 $O/r06_${TAG}_bench.json; echo; tail -12 $O/r06_${TAG}_kernel_stats.txt; cat $O/r06_${TAG}_pmc_hbm_traffic.txt $O/r06_${TAG}_pmc_mfma_util.txt
