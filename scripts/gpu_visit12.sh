#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_l; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -n 8 $O/pytest.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h = d.get("h2d_inclusive") or {}
    print(sys.argv[2], round(d["ms_per_step"] * 1e3, 2), d.get("match_vs_oracle"), {k: round(v["avg_us"], 2) for k, v in d.get("kernels", {}).items()}, (d.get("roofline") or {}).get("frac"), h.get("ms_per_step"))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for w in c2 c2n c2e c2k3 c2d c3 c1 sd; do timeout 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; show $O/bench_$w.json "$w"; done
python scripts/bench_tracker.py 1000 512 30 2>/dev/null | tail -4
echo DONE
