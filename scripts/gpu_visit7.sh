#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_g; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -n 6 $O/pytest.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h = d.get("h2d_inclusive") or {}
    print(sys.argv[2], {k: d.get(k) for k in ("ms_per_step", "value_h2d")}, {k: h.get(k) for k in ("ms_per_step", "synchronous_ms_per_step", "host_us_in_submit", "host_us_in_wait", "tickets_in_flight")})
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for ev in attach record; do for b in 24 32; do
  SA_INGEST_EVENT=$ev SA_INGEST_BLOCKS=$b timeout 300 python bench.py --no-cpu-baseline --no-oracle > $O/bench_c2_${ev}_b$b.json 2> $O/bench_c2_${ev}_b$b.err; show $O/bench_c2_${ev}_b$b.json "c2 event=$ev blocks=$b"
done; done
echo DONE
