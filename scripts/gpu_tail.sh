#!/bin/bash
# the assignment tails: risky kernels first under short timeouts, then the suite, then the bench lines that exercise them
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r03_t}; O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; }
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -rf -k "one_giant or crowds or dense_positional or big_frames or big_visual or general_assignment or full_size_c4" > $O/pytest_tail.log 2>&1; echo "tail tests exit $?"; grep -E "^FAILED|passed|failed" $O/pytest_tail.log | cut -c1-300 | tail -12
if [ "$2" != "quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --maxfail=40 -rf > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; grep -E "^(FAILED|ERROR)|passed|failed|exit" $O/pytest.log | cut -c1-300 | tail -n 30
fi
for w in c4 c5 bigcrowd bigpile giant sd c3; do
  st=""; [ "$w" = "c5" ] && st="--steps 20 --warmup 3 --profile-iters 10"
  timeout 400 python bench.py --workload $w --no-cpu-baseline --no-h2d $st > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w exit $?"
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
