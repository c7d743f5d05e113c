#!/bin/bash
# In-kernel timelines on the GPU box: rebuilds the library with the trace hooks compiled in, runs the positional-tile timeline
# (scripts/pos_trace.sh) and the contraction timeline (scripts/gemm_trace.sh) for the workloads given, plus the launch-floor micro-benchmark.
#   scripts/gpu_trace.sh <tag> "<pos workloads>" "<gemm workloads>"
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r04_t}; POSW=${2:-c4 c3}; GEMW=${3:-c2t}
O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
if [ -z "$SKIP_FLOOR" ] && [ -x scripts/micro/launch_floor.bin ]; then scripts/micro/launch_floor.bin > $O/launch_floor.txt 2>&1; cat $O/launch_floor.txt; fi
SA_EXTRA_FLAGS="-DSA_POS_TRACE -DSA_GEMM_TRACE" python -m similari_amd.build --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; exit 1; }
WORKLOADS="$POSW" bash scripts/pos_trace.sh > $O/pos_trace.txt 2>&1; cat $O/pos_trace.txt
for w in $GEMW; do
  echo "== gemm trace $w" | tee -a $O/gemm_trace.txt
  WORKLOAD=$w bash scripts/gemm_trace.sh 0 2>&1 | tee -a $O/gemm_trace.txt
  cp gpurun_out/gemm_trace.txt $O/gemm_trace_$w.raw 2>/dev/null
done
echo DONE
