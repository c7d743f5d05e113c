#!/bin/bash
# Timeline of k_assign_solve (library built with SA_EXTRA_FLAGS=-DSA_TAIL_TRACE; the build on the box takes ~2 min) -> gpurun_out/solve_trace.txt
#   scripts/solve_trace.sh sdt c4 ...      (bench workloads)   scripts/solve_trace.sh tracker:sort,rows,0.0
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; rm -f gpurun_out/solve_trace.txt
# (SA_TRACE_PREBUILT=1: the library in the tree was built with the flag already — e.g. before a gpurun call, to spare the box the build)
[ -n "$SA_TRACE_PREBUILT" ] || SA_EXTRA_FLAGS="-DSA_TAIL_TRACE" python -m similari_amd.build --force > gpurun_out/build_trace.log 2>&1 || { echo BUILD FAILED; tail -20 gpurun_out/build_trace.log; exit 1; }
for w in "$@"; do
  echo "== $w" >> gpurun_out/solve_trace.txt
  case "$w" in
    tracker:*) SA_SOLVE_TRACE=25 SA_BENCH_TRACKER_ONLY="${w#tracker:}" timeout 300 python scripts/bench_tracker.py 1000 512 30 > /dev/null 2> gpurun_out/st.err ;;
    *) SA_SOLVE_TRACE=30 timeout 300 python bench.py --workload $w --no-cpu-baseline --no-oracle --no-h2d --steps 40 --warmup 5 > /dev/null 2> gpurun_out/st.err ;;
  esac
done
cat gpurun_out/solve_trace.txt
[ -n "$SA_TRACE_PREBUILT" ] || python -m similari_amd.build --force > /dev/null 2>&1
