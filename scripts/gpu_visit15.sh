#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_o; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "general_assignment_tail or full_size_c4 or full_size_sort" > $O/pytest_first.log 2>&1; echo "first exit $?" >> $O/pytest_first.log; tail -5 $O/pytest_first.log
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider --durations=5 -k "big_frames or big_visual" > $O/pytest_big.log 2>&1; echo "big exit $?" >> $O/pytest_big.log; tail -14 $O/pytest_big.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -n 6 $O/pytest.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["ms_per_step"] * 1e3, 2), d.get("match_vs_oracle"), {k: round(v["avg_us"], 2) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for w in c4 c5; do timeout 600 python bench.py --workload $w --no-cpu-baseline --no-h2d > $O/bench_$w.json 2> $O/bench_$w.err; show $O/bench_$w.json "$w"; done
echo DONE
