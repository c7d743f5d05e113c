#!/bin/bash
# A/B of the fused first phase's main loop: k-split over the fragment-order bank (default) against the LDS-staged loop (SA_FLAG_STAGED_LOOP)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
for wl in ${WORKLOADS:-c2 c2t c2e c2k3}; do
  for fl in 0 131072; do
    timeout 300 python bench.py --workload $wl --no-cpu-baseline --flags $fl > gpurun_out/ks_ab_${wl}_$fl.json 2> gpurun_out/ks_ab_${wl}_$fl.err || tail -5 gpurun_out/ks_ab_${wl}_$fl.err
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ks_ab_${wl}_$fl.json").read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print("$wl flags $fl: ms_per_step %.5f value %.3g match %s/%s kernel %s avg_us %s frac %s" % (d["ms_per_step"], d["value"], d.get("match_vs_oracle"), d.get("match_accuracy"), r.get("kernel"), r.get("avg_us"), r.get("frac")))
except Exception as ex: print("$wl $fl failed", ex)
PY
  done
done
