#!/bin/bash
# One GPU-box visit of round 3: build, the GPU parity suite (all failures listed, not only the first), smoke, optional bench lines.
#   scripts/gpu_visit.sh <tag> [pytest -k expression | all] [bench workloads ...]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r03_a}; KEXPR=${2:-all}; shift; shift
O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
(rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6; nproc) > $O/env.txt 2>&1
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; }
if [ "$KEXPR" = "all" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --maxfail=40 -rf > $O/pytest.log 2>&1
elif [ "$KEXPR" != "none" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --maxfail=40 -rf -k "$KEXPR" > $O/pytest.log 2>&1
fi
echo "pytest exit $?" >> $O/pytest.log; grep -E "^(FAILED|ERROR)|passed|failed|exit" $O/pytest.log | cut -c1-400 | tail -n 60
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log; tail -n 2 $O/smoke.log
for w in "$@"; do
  if [ "$w" = "c2" ]; then timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
  else timeout 400 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; fi
  echo "bench $w exit $?"
  python - "$O/bench_$w.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "value_h2d", "match_accuracy", "match_vs_oracle")}, {k: round(v["avg_us"], 2) for k, v in d.get("kernels", {}).items()},
          (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print("unreadable", e)
PY
done
echo DONE
