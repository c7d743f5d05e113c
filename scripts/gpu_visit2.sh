#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_b; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
hipcc --offload-arch=gfx950 -O3 scripts/micro/h2d_rate.hip -o /tmp/h2d_rate > $O/h2d_build.log 2>&1 && timeout 120 /tmp/h2d_rate > $O/h2d_rate.txt 2>&1; cat $O/h2d_rate.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -n 30 $O/pytest.log
for a in 256 4096 65536 2097152; do
  SA_FEAT_ALIGN=$a timeout 300 python bench.py --workload c5 --no-cpu-baseline --no-oracle --no-h2d --steps 50 --warmup 5 --profile-iters 20 > $O/bench_c5_a$a.json 2> $O/bench_c5_a$a.err
  python - $O/bench_c5_a$a.json $a <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("c5 align", sys.argv[2], d["ms_per_step"], {k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
PY
done
timeout 300 python bench.py --no-cpu-baseline --no-oracle > $O/bench_c2.json 2> $O/bench_c2.err; python - $O/bench_c2.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"], d["h2d_inclusive"])
PY
echo DONE
