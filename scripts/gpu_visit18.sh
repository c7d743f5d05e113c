#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_s; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["ms_per_step"] * 1e3, 2), round(d["timed_regions"]["ms_per_step_slope"] * 1e3, 2), d.get("match_vs_oracle"), {k: round(v["avg_us"], 2) for k, v in d.get("kernels", {}).items()}, (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for t in 0 1 2 0 1 2; do SA_FUSED_POS=$t timeout 300 python bench.py --no-cpu-baseline --no-h2d > $O/bench_c2_t$t.json 2> $O/bench_c2_t$t.err; show $O/bench_c2_t$t.json "c2 fused_pos=$t"; done
for t in 0 1 2; do SA_FUSED_POS=$t timeout 300 python bench.py --workload c2n --no-cpu-baseline --no-h2d > $O/bench_c2n_t$t.json 2> $O/bench_c2n_t$t.err; show $O/bench_c2n_t$t.json "c2n fused_pos=$t"; done
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > $O/bench_c2_driver.json 2> $O/bench_c2_driver.err; show $O/bench_c2_driver.json "c2 driver command"
echo DONE
