#!/bin/bash
# Two-rank rehearsal on the one-GPU box: the cluster tests, then bench.py --gpus 2 on the full 64-scene sets (both ranks drive device 0, gloo).
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r05_ranks}; O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; }
timeout 1200 python -m pytest tests/test_gpu_cluster.py -m gpu -q --timeout 900 -p no:cacheprovider --maxfail=5 -rf > $O/pytest_cluster.log 2>&1; echo "cluster exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $O/pytest_cluster.log | cut -c1-300 | tail -15
for w in c2 c3; do
  SA_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --workload $w --steps 20 --warmup 5 --profile-iters 5 --no-cpu-baseline --no-oracle --no-h2d > $O/bench_2ranks_$w.json 2> $O/bench_2ranks_$w.err
  echo "bench 2 ranks $w exit $?"
  python - $O/bench_2ranks_$w.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "value_h2d", "value_scatter", "value_resident", "ms_per_step_resident")})
    print(json.dumps(d.get("ingest_local"), indent=0)[:1500])
    print(json.dumps(d.get("c3_batchsort"), indent=0)[:900])
except Exception as e:
    print("unreadable", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-2000:])
PY
done
echo DONE
