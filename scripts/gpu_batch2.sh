#!/bin/bash
# Batch facade experiments: per-phase medians of SA_TRACKER_TRACE for several pool sizes, pinned or not.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r05_c}; O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; }
g++ -O2 -std=c++17 -pthread scripts/micro/pool_bench.cpp -o /tmp/pool_bench && /tmp/pool_bench
agg() {
python - "$1" <<'PY'
import re, sys, statistics as st
L = [l for l in open(sys.argv[1]) if l.startswith("[sa_tracker]")]
for key in ("up to the launches", "behind the launches"):
    rows = [[float(x) for x in re.findall(r"[-+]?\d+\.\d+", l)] for l in L if key in l][4:]
    if rows:
        print("   ", key, [round(st.median(c), 1) for c in zip(*rows)])
PY
}
for cfg in "sort 64 500 0 40 0" "sort 64 500 0 40 8" "sort 8 500 0 40 0" "sort 8 500 0 40 4" "visual 8 1000 512 24 0"; do
  echo "== $cfg"
  timeout 200 python scripts/bench_batch_tracker.py $cfg sync device | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ', d['us_per_predict_median'], d['us_per_predict_min'])"
  SA_TRACKER_TRACE=1 timeout 200 python scripts/bench_batch_tracker.py $cfg sync device 2> $O/tr.txt > /dev/null; agg $O/tr.txt
done
echo DONE
