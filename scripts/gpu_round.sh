#!/bin/bash
# One GPU-box visit of round 2+: risky-kernel smoke first (short timeouts), the GPU parity suite, the bench lines of every workload,
# rocprofv3 kernel stats of the default bench command.   scripts/gpu_round.sh <tag> [quick]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r02_a}; MODE=${2:-full}; O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
(rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6; nproc) > $O/env.txt 2>&1
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; }
# 1. the new assignment tail on small cases, each under its own short timeout: a spinning kernel must not eat the visit
timeout 180 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "test_sort_iou_parity or test_sort_maha_parity" > $O/pytest_first.log 2>&1; echo "first exit $?" >> $O/pytest_first.log; tail -3 $O/pytest_first.log
timeout 240 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "giant or crowds or dense_positional" > $O/pytest_crowd.log 2>&1; echo "crowd exit $?" >> $O/pytest_crowd.log; tail -5 $O/pytest_crowd.log
# 2. everything
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -n 25 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log; tail -n 2 $O/smoke.log
# 3. bench lines
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 exit $?"; cut -c1-1500 $O/bench_c2.json
for w in c2n c2e c2k3 c3 c3m c4 c1 sd giant; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w exit $?"
  python - "$O/bench_$w.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "value_h2d", "match_accuracy", "match_vs_oracle")}, {k: round(v["avg_us"], 2) for k, v in d.get("kernels", {}).items()},
          (d.get("h2d_inclusive") or {}).get("ms_per_step"), (d.get("h2d_inclusive") or {}).get("synchronous_ms_per_step"))
except Exception as e:
    print("unreadable", e)
PY
done
if [ "$MODE" = "full" ]; then
  timeout 600 python bench.py --workload c5 --no-cpu-baseline --steps 20 --warmup 3 --profile-iters 10 > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 exit $?"; cut -c1-600 $O/bench_c5.json
  timeout 300 python bench.py --flags 32 --no-cpu-baseline > $O/bench_c2_separate.json 2> $O/bench_c2_separate.err
  SA_COOP_G=16 timeout 300 python bench.py --workload giant --no-cpu-baseline > $O/bench_giant_g16.json 2> $O/bench_giant_g16.err
  SA_COOP_G=16 timeout 300 python bench.py --workload sd --no-cpu-baseline > $O/bench_sd_g16.json 2> $O/bench_sd_g16.err
  timeout 300 python bench.py --workload c3 --no-cpu-baseline --cluster 2 --cluster-devices 0,0 > $O/bench_c3_cluster2.json 2> $O/bench_c3_cluster2.err; echo "cluster exit $?"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o bench -- python $OLDPWD/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-oracle > $O/prof_c2.log 2>&1)
  f=$(find $O/prof_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c2_kernel_stats.csv && head -8 $O/c2_kernel_stats.csv
  find $O/prof_c2 -name "*kernel_trace.csv" -size +4M -delete
fi
echo DONE
