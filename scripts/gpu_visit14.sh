#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_n; rm -rf $O; mkdir -p $O
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
for pp in 0 1 3; do for gp in 0 1 3; do
  SA_POS_PRIO=$pp SA_GEMM_PRIO=$gp timeout 300 python bench.py --no-cpu-baseline --no-oracle --no-h2d > $O/bench_c2_p${pp}_g$gp.json 2> $O/bench_c2_p${pp}_g$gp.err; show $O/bench_c2_p${pp}_g$gp.json "c2 pos_prio=$pp gemm_prio=$gp"
done; done
echo DONE
