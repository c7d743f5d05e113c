#!/bin/bash
# The whole GPU suite, then the default bench line.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r05_full}; O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=5 -rf > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-250 | tail -12
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench exit $?"
python - $O/bench_c2.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "value_h2d")}, "device_features", d.get("device_features_inclusive", {}).get("pairs_per_s"), "roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "frac_rocprof")}, "match", d.get("match_vs_oracle"))
PY
