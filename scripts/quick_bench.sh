#!/bin/bash
# GPU parity tests (optionally a subset via PYTEST_ARGS) and the five bench workloads, kernel table only.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider ${PYTEST_ARGS:-} 2>&1 | tail -8
for w in ${WORKLOADS:-c2 c3 c3m c4 c5}; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline ${BENCH_ARGS:-} 2> gpurun_out/qb_$w.err > gpurun_out/qb_$w.json
  python - <<PY
import json
d=json.load(open("gpurun_out/qb_$w.json"))
print("$w", round(d["ms_per_step"]*1000,2), "us/step acc", d.get("match_accuracy"), {k:round(v["avg_us"],2) for k,v in d["kernels"].items()})
PY
done
