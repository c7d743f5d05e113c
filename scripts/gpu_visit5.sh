#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_e; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -n 8 $O/pytest.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h = d.get("h2d_inclusive") or {}
    print(sys.argv[2], {k: d.get(k) for k in ("ms_per_step", "value_h2d", "match_vs_oracle")}, {k: (round(v["avg_us"], 2), round(v["avg_us_instrumented"], 2)) for k, v in d.get("kernels", {}).items()},
          {k: h.get(k) for k in ("ms_per_step", "synchronous_ms_per_step", "host_us_in_submit", "host_us_in_wait")}, (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for p in 1 0; do for b in 24 48; do SA_COPY_PRIO=$p SA_INGEST_BLOCKS=$b timeout 300 python bench.py --no-cpu-baseline --no-oracle > $O/bench_c2_p${p}_b$b.json 2> $O/bench_c2_p${p}_b$b.err; show $O/bench_c2_p${p}_b$b.json "c2 prio=$p blocks=$b"; done; done
timeout 300 python bench.py --workload c2k3 --no-cpu-baseline > $O/bench_c2k3.json 2> $O/bench_c2k3.err; show $O/bench_c2k3.json "c2k3"
echo DONE
