#!/bin/bash
# GPU visit for the dense assignment solver: the risky kernels first under short timeouts, then the suite, then the bench lines.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r03_b}; O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; }
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "test_one_giant_component" > $O/pytest_giant.log 2>&1; echo "giant exit $?"; tail -4 $O/pytest_giant.log | cut -c1-600
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -rf -k "crowds or dense_positional or big_frames or big_visual or general_assignment" > $O/pytest_crowd.log 2>&1; echo "crowd exit $?"; grep -E "^FAILED|passed|failed" $O/pytest_crowd.log | cut -c1-300 | tail -12
if [ "$2" != "quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --maxfail=40 -rf > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; grep -E "^(FAILED|ERROR)|passed|failed|exit" $O/pytest.log | cut -c1-300 | tail -n 40
fi
for w in giant bigpile bigcrowd sd c4 c3 c2; do
  if [ "$w" = "c2" ]; then timeout 600 python bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
  else timeout 400 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; fi
  echo "bench $w exit $?"
  python - "$O/bench_$w.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "match_accuracy", "match_vs_oracle")}, {k: round(v["avg_us"], 2) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print("unreadable", e)
PY
done
echo DONE
