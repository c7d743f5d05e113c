#!/bin/bash
# Timeline of k_assign_small (library built with SA_EXTRA_FLAGS=-DSA_TAIL_TRACE) -> gpurun_out/tail_trace.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; rm -f gpurun_out/tail_trace.txt
for w in ${WORKLOADS:-c2 c3 c3m}; do
  echo "== $w" >> gpurun_out/tail_trace.txt
  SA_TAIL_TRACE=30 timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 40 --warmup 5 > /dev/null 2> gpurun_out/tt_$w.err
done
cat gpurun_out/tail_trace.txt
