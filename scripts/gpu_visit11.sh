#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_k; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -n 6 $O/pytest.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["ms_per_step"] * 1e3, 2), d.get("match_vs_oracle"), {k: round(v["avg_us"], 2) for k, v in d.get("kernels", {}).items()}, (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
timeout 300 python bench.py --workload c2d --no-cpu-baseline --no-h2d > $O/bench_c2d.json 2> $O/bench_c2d.err; show $O/bench_c2d.json "c2d"
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-h2d > $O/bench_c2_$i.json 2> $O/bench_c2_$i.err; show $O/bench_c2_$i.json "c2 run $i"
SA_SKIP_PREP=1 timeout 300 python bench.py --no-cpu-baseline --no-h2d > $O/bench_c2_noprep_$i.json 2> $O/bench_c2_noprep_$i.err; show $O/bench_c2_noprep_$i.json "c2 no prep run $i"
done
echo DONE
