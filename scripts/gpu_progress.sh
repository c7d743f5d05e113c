#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r05_prog}; O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; }
timeout 1200 python -m pytest tests/test_gpu_progress.py -m gpu -q -s --timeout 1000 -p no:cacheprovider -rs > $O/pytest_progress.log 2>&1; echo "progress exit $?"; tail -15 $O/pytest_progress.log | cut -c1-600
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider --maxfail=10 -rf -k "crowd or general or big or tail or beyond or refusal or 1024 or mahalanobis" > $O/pytest_tail.log 2>&1; echo "tail exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_tail.log | cut -c1-300 | tail -12
for w in sdt c4 bigcrowd c2t; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --no-h2d > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w exit $?"
  python - "$O/bench_$w.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "match_vs_oracle")}, {k: round(v["avg_us"], 2) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print("unreadable", e)
PY
done
echo DONE
