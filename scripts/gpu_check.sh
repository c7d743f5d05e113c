#!/bin/bash
# One GPU-box visit: build check, GPU parity tests, smoke, bench, rocprof kernel trace. Outputs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo ==" > gpurun_out/env.log
(rocminfo | grep -E "Name:|Compute Unit|Max Clock" | head -20; nproc; rocm-smi --showmeminfo vram 2>/dev/null | head -8) >> gpurun_out/env.log 2>&1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -20 gpurun_out/build.log; }
WHAT="${1:-all}"
if [ "$WHAT" = "all" ] || [ "$WHAT" = "test" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest.log 2>&1
  echo "pytest exit $?" >> gpurun_out/pytest.log
  tail -n 60 gpurun_out/pytest.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
  tail -n 5 gpurun_out/smoke.log
fi
if [ "$WHAT" = "all" ] || [ "$WHAT" = "bench" ]; then
  timeout 600 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"
  tail -n 3 gpurun_out/bench.err; cat gpurun_out/bench.json
  for w in ${EXTRA_WORKLOADS:-}; do
    timeout 600 python bench.py --workload $w --no-cpu-baseline > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err; echo "bench $w exit $?"
    cat gpurun_out/bench_$w.json
  done
fi
if [ "$WHAT" = "all" ] || [ "$WHAT" = "prof" ]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 50 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} > "$OLDPWD/gpurun_out/prof_run.log" 2>&1; echo "rocprof exit $?")
  find gpurun_out/prof -name "*stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
  # keep the merge-back small: drop the big per-dispatch traces
  find gpurun_out/prof -name "*kernel_trace.csv" -size +8M -delete
fi
