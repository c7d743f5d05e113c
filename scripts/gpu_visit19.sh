#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_u; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "euclid" > $O/pytest_first.log 2>&1; echo "first exit $?" >> $O/pytest_first.log; tail -4 $O/pytest_first.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -n 5 $O/pytest.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["ms_per_step"] * 1e3, 2), d.get("match_vs_oracle"), {k: round(v["avg_us"], 2) for k, v in d.get("kernels", {}).items()}, (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for w in c2e c2d c2; do timeout 300 python bench.py --workload $w --no-cpu-baseline --no-h2d > $O/bench_$w.json 2> $O/bench_$w.err; show $O/bench_$w.json "$w"; done
echo DONE
