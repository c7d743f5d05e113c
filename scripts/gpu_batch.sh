#!/bin/bash
# One GPU-box visit for the batch facade: build, the tracker tests, Batch*::predict timings (facade vs device timeline).
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r05_a}; O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
(rocminfo | grep -E "Marketing Name|Compute Unit" | head -4; nproc) > $O/env.txt 2>&1
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; }
timeout 900 python -m pytest tests/test_trackers.py tests/test_gpu_pipeline.py tests/test_gpu_cluster.py -m gpu -q --timeout 300 -p no:cacheprovider --maxfail=15 -rf > $O/pytest_trk.log 2>&1
echo "trk exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_trk.log | cut -c1-300 | tail -20
J=$O/batch_tracker.jsonl; : > $J
run() { timeout 300 python scripts/bench_batch_tracker.py "$@" >> $J 2>> $O/batch_err.txt || echo "bench_batch_tracker $* failed"; }
run sort 8 500 0 60 0 sync
run sort 8 500 0 60 1 sync
run sort 8 500 0 60 4 sync
run sort 8 500 0 60 0 async
run sort 64 500 0 60 0 sync
run sort 64 500 0 60 1 sync
run sort 64 500 0 60 8 sync
run sort 64 500 0 60 0 async
run visual 8 1000 512 24 0 sync device
run visual 8 1000 512 24 1 sync device
run visual 8 1000 512 24 0 sync rows
run visual 8 1000 512 24 0 async device
cat $J
for cfg in "sort 64 500 0 40 0" "sort 8 500 0 40 0" "visual 8 1000 512 24 0"; do
  tag=$(echo $cfg | tr ' ' '_')
  SA_TRACKER_TRACE=1 timeout 200 python scripts/bench_batch_tracker.py $cfg sync device 2> $O/trace_$tag.txt > /dev/null
  python - $O/trace_$tag.txt "$cfg" <<'PY' | tee -a $O/batch_tracker_phases.txt
import re, sys, statistics as st
L = [l for l in open(sys.argv[1]) if l.startswith("[sa_tracker]")]
print(sys.argv[2], "(medians over the calls, us)")
names = {"up to the launches": ["before the jobs", "assemble", "longest assemble job", "epochs", "stage + evict", "enqueue"],
         "behind the launches": ["deferred", "wait for the association", "merges", "longest merge job", "all merge jobs", "wait for the Kalman dispatch",
                                 "tables + results", "longest job", "all jobs", "minor faults"]}
for key, nm in names.items():
    rows = [[float(x) for x in re.findall(r"[-+]?\d+\.\d+", l)] for l in L if key in l][4:]
    if rows:
        print("   ", key + ":", dict(zip(nm, [round(st.median(c), 1) for c in zip(*rows)])))
PY
done
scripts/batch_tracker_timeline.sh ${TAG}_s8 sort 8 500 0 40 0 sync > $O/timeline_sort8.txt 2>&1; cat $O/timeline_sort8.txt
scripts/batch_tracker_timeline.sh ${TAG}_s64 sort 64 500 0 24 0 sync > $O/timeline_sort64.txt 2>&1; cat $O/timeline_sort64.txt
scripts/batch_tracker_timeline.sh ${TAG}_v8 visual 8 1000 512 24 0 sync device > $O/timeline_visual8.txt 2>&1; cat $O/timeline_visual8.txt
SA_BENCH_TRACKER_ONLY="sort,rows,0.0" timeout 200 python scripts/bench_tracker.py 1000 512 30; SA_BENCH_TRACKER_ONLY="visual,device,0.0" timeout 200 python scripts/bench_tracker.py 1000 512 30
SA_BENCH_TRACKER_ONLY="sort,rows,0.05" timeout 200 python scripts/bench_tracker.py 1000 512 30; SA_BENCH_TRACKER_ONLY="visual,device,0.05" timeout 200 python scripts/bench_tracker.py 1000 512 30
echo DONE
g++ -O2 -std=c++17 -pthread scripts/micro/pool_bench.cpp -o /tmp/pool_bench && /tmp/pool_bench
