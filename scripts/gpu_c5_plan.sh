#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
for wl in c5 c2b; do
for plan in 16 18 5 18 16; do
  st="--steps 20 --warmup 3 --profile-iters 10"; [ "$wl" = "c2b" ] && st="--steps 50 --warmup 5 --profile-iters 10"
  timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-h2d $st --gemm-plan $plan > gpurun_out/c5p.json 2> gpurun_out/c5p.err || tail -3 gpurun_out/c5p.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/c5p.json").read().strip().splitlines()[-1])
print("$wl plan $plan: ms_per_step %.5f match %s frac %.4f" % (d["ms_per_step"], d.get("match_vs_oracle"), d["roofline"]["frac"]))
PY
done
done
