#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_i; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: d.get(k) for k in ("ms_per_step", "match_vs_oracle")}, {k: (round(v["avg_us"], 2), round(v["avg_us_instrumented"], 2)) for k, v in d.get("kernels", {}).items()}, (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for kg in 1 2; do for w in c2 c2n; do SA_FRAME_KG=$kg timeout 300 python bench.py --workload $w --no-cpu-baseline --no-h2d > $O/bench_${w}_kg$kg.json 2> $O/bench_${w}_kg$kg.err; show $O/bench_${w}_kg$kg.json "$w frame_kg=$kg"; done; done
echo DONE
