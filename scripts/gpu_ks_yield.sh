#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
for wl in ${WORKLOADS:-c2}; do
  for y in ${YIELDS:-0 1 2 4 8}; do
    SA_KS_YIELD=$y timeout 300 python bench.py --workload $wl --no-cpu-baseline > gpurun_out/ks_y.json 2> gpurun_out/ks_y.err || tail -5 gpurun_out/ks_y.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/ks_y.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("$wl yield $y: ms_per_step %.5f match %s frac %.4f" % (d["ms_per_step"], d.get("match_vs_oracle"), r.get("frac")))
PY
  done
done
