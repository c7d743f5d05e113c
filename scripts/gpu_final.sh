#!/bin/bash
# Last visit of a round on the final tree: the whole GPU suite, smoke, the driver's default bench command (and its --steps 20 --warmup 5 form),
# the bench lines whose roofline reads this round's committed profile set, the two-rank rehearsal on one device.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r06_z}; O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; }
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --maxfail=20 -rf > $O/pytest.log 2>&1; echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | cut -c1-300 | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -n 1 $O/smoke.log
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench default exit $?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver_command.json 2> $O/bench_c2_driver_command.err; echo "bench driver command exit $?"
for w in c2e c2t c2k3 c2b c5 c4 c3 vref c1ref_or; do
  st=""; [ "$w" = "c5" ] && st="--steps 20 --warmup 3 --profile-iters 10"; [ "$w" = "c2b" ] && st="--steps 50 --warmup 5 --profile-iters 10"
  timeout 600 python bench.py --workload $w --no-cpu-baseline $st > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w exit $?"
done
for w in c2 c3; do
  SA_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --workload $w --steps 20 --warmup 5 --profile-iters 5 --no-cpu-baseline --no-oracle --no-h2d > $O/bench_2ranks_$w.json 2> $O/bench_2ranks_$w.err
  echo "bench 2 ranks $w exit $?"
done
python - $O <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d.get("roofline") or {}
        print(os.path.basename(f), {k: d.get(k) for k in ("value", "ms_per_step", "value_h2d", "value_scatter", "value_resident")}, "frac", r.get("frac"), "rocprof", r.get("frac_rocprof"), "traffic", r.get("traffic"), r.get("traffic_stale"))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
echo DONE
