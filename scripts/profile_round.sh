#!/bin/bash
# Round profile collection on the GPU box: rocprofv3 kernel stats of the default bench command (and of the other headline workloads),
# bench lines of every workload, HBM-traffic PMC passes (separate --pmc runs with --kernel-trace only), MFMA utilisation counters, the
# tracker loop, and — last, because they rebuild the library with the trace hooks — the in-kernel timelines.
#   scripts/profile_round.sh <tag>        outputs under gpurun_out/<tag>/ — copy what is to be judged into profiles/
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r05_z}; O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
(rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6; nproc) > $O/env.txt 2>&1
# 1. kernel stats: the DEFAULT command first (the driver's line), then the other workloads
for w in c2 c2b c2t c2e c5 c4 c3 c2k3 sdt; do
  extra="--workload $w --no-cpu-baseline --no-oracle --no-h2d"; [ "$w" = "c2" ] && extra="--no-cpu-baseline --no-oracle --no-h2d"
  steps="--steps 50 --warmup 5"; [ "$w" = "c5" ] && steps="--steps 10 --warmup 2 --profile-iters 5"; [ "$w" = "c2b" ] && steps="--steps 20 --warmup 3 --profile-iters 10"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o bench -- python $OLDPWD/bench.py $steps $extra > $O/prof_$w.log 2>&1)
  f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${w}_kernel_stats.csv && echo "== $w" && head -5 $O/${w}_kernel_stats.csv | cut -c1-160
  rm -rf $O/prof_$w
done
# 2. bench lines (the default one with its CPU baselines)
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 exit $?"
for w in c2b c2bk3 c2t c2n c2e c2k3 c2d c3 c3m c4 c5 c1 c1ref sd sdt giant bigpile bigcrowd; do
  st=""; [ "$w" = "c5" ] && st="--steps 20 --warmup 3 --profile-iters 10"; [ "$w" = "c2b" -o "$w" = "c2bk3" ] && st="--steps 50 --warmup 5 --profile-iters 10"
  timeout 600 python bench.py --workload $w --no-cpu-baseline $st > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w exit $?"
done
timeout 300 python bench.py --flags 32 --no-cpu-baseline > $O/bench_c2_separate.json 2> $O/bench_c2_separate.err
timeout 300 python bench.py --flags 32768 --no-cpu-baseline --no-oracle --no-h2d > $O/bench_c2_row_tiles.json 2> $O/bench_c2_row_tiles.err
timeout 300 python bench.py --workload c2b --gemm-plan 1 --steps 50 --warmup 5 --profile-iters 10 --no-cpu-baseline --no-oracle --no-h2d > $O/bench_c2b_fused64.json 2> $O/bench_c2b_fused64.err
# 3. HBM traffic per launch (FETCH_SIZE / WRITE_SIZE in separate passes); C2 also with its tiles row by row (SA_FLAG_ROW_TILES)
for w in c2 c2b c5 c4 c2e c2t c2k3; do bash scripts/pmc_traffic.sh $w ${TAG}_$w > $O/pmc_traffic_$w.log 2>&1; cp gpurun_out/pmc_${TAG}_$w/summary.json $O/pmc_traffic_$w.json 2>/dev/null; done
SA_BENCH_FLAGS=32768 bash scripts/pmc_traffic.sh c2 ${TAG}_c2row > $O/pmc_traffic_c2_row_tiles.log 2>&1; cp gpurun_out/pmc_${TAG}_c2row/summary.json $O/pmc_traffic_c2_row_tiles.json 2>/dev/null
# 4. MFMA utilisation of the contraction (C2 default line, c2b and C5)
for w in c2 c2b c5; do
  extra="--workload $w"; st="--steps 30 --warmup 5"; [ "$w" = "c5" ] && st="--steps 8 --warmup 2 --profile-iters 4"; [ "$w" = "c2b" ] && st="--steps 10 --warmup 2 --profile-iters 4"
  (cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_$w -o p -- \
     python $OLDPWD/bench.py $extra $st --no-cpu-baseline --no-oracle --no-h2d > $O/pmc_mfma_$w.log 2>&1)
  python - "$O/pmc_mfma_$w" "$O/mfma_util_$w.json" <<'PY'
import csv, glob, json, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        m["launches"] = len(next(iter(d.values())))
        if m.get("SQ_BUSY_CYCLES"): m["mfma_busy_over_sq_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CYCLES"]
        out[k] = m
json.dump(out, open(sys.argv[2], "w"), indent=1)
print({k: round(v.get("mfma_busy_over_sq_busy", 0), 3) for k, v in out.items()})
PY
  rm -rf $O/pmc_mfma_$w
done
# 5. the tracker loop (plain and churned) and where a predict() spends its time
timeout 600 python scripts/bench_tracker.py 1000 512 30 > $O/tracker_loop.jsonl 2> $O/tracker_loop.err; cat $O/tracker_loop.jsonl
bash scripts/tracker_trace.sh "visual,device,0.0" "visual,pinned,0.0" "sort,rows,0.0" "visual,device,0.05" "sort,rows,0.05" > $O/tracker_breakdown.txt 2>&1; cat $O/tracker_breakdown.txt
# 5b. Batch*::predict through the facade: per call, per phase, device timelines (scripts/gpu_batch.sh without its tests)
J=$O/batch_tracker.jsonl; : > $J
brun() { timeout 300 python scripts/bench_batch_tracker.py "$@" >> $J 2>> $O/batch_err.txt || echo "bench_batch_tracker $* failed"; }
brun sort 8 500 0 60 0 sync; brun sort 8 500 0 60 1 sync; brun sort 8 500 0 60 0 async
brun sort 64 500 0 60 0 sync; brun sort 64 500 0 60 1 sync; brun sort 64 500 0 60 8 sync; brun sort 64 500 0 60 0 async
brun visual 8 1000 512 24 0 sync device; brun visual 8 1000 512 24 1 sync device; brun visual 8 1000 512 24 0 sync rows; brun visual 8 1000 512 24 0 async device
cat $J
scripts/batch_tracker_timeline.sh ${TAG}_s8 sort 8 500 0 40 0 sync > $O/batch_tracker_timeline_sort8.txt 2>&1
scripts/batch_tracker_timeline.sh ${TAG}_s64 sort 64 500 0 24 0 sync > $O/batch_tracker_timeline_sort64.txt 2>&1
scripts/batch_tracker_timeline.sh ${TAG}_v8 visual 8 1000 512 24 0 sync device > $O/batch_tracker_timeline_visual8.txt 2>&1
tail -n 8 $O/batch_tracker_timeline_*.txt
bash scripts/tracker_timeline.sh "visual,device,0.0" > $O/tracker_timeline_visual.txt 2>&1; bash scripts/tracker_timeline.sh "sort,None,0.0" > $O/tracker_timeline_sort.txt 2>&1
tail -n 7 $O/tracker_timeline_*.txt
[ -n "$SA_PROFILE_SKIP_TRACES" ] && { echo "DONE (without the in-kernel timelines)"; exit 0; }
# 5c. the giant components: where a search step's time goes (k_assign_solve's timeline; rebuilds with -DSA_TAIL_TRACE)
bash scripts/solve_trace.sh giant bigcrowd sdt > $O/solve_trace.txt 2>&1; tail -30 $O/solve_trace.txt
# 6. in-kernel timelines (rebuilds the library with -DSA_POS_TRACE -DSA_GEMM_TRACE) + the launch floor
bash scripts/gpu_trace.sh ${TAG}_trace "c4 c3 c1" "c2 c2t" > $O/trace.log 2>&1; cp gpurun_out/${TAG}_trace/*.txt $O/ 2>/dev/null; tail -45 $O/trace.log
echo DONE
