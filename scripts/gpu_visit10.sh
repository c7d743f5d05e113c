#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_j; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["ms_per_step"] * 1e3, 2), {k: round(v["avg_us"], 2) for k, v in d.get("kernels", {}).items()}, (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for w in c2d c2k3; do for plan in default 0 5 6 1 2; do
  if [ $plan = default ]; then unset SA_GEMM_PLAN; else export SA_GEMM_PLAN=$plan; fi
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-oracle --no-h2d --steps 100 > $O/bench_${w}_p$plan.json 2> $O/bench_${w}_p$plan.err; show $O/bench_${w}_p$plan.json "$w plan=$plan"
done; done
unset SA_GEMM_PLAN
echo DONE
