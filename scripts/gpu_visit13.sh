#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_m; rm -rf $O; mkdir -p $O
SA_EXTRA_FLAGS=-DSA_GEMM_TRACE python -m similari_amd.build --force > $O/build.log 2>&1
bash scripts/gemm_trace.sh 0 32 > $O/gemm_trace.txt 2>&1; cat $O/gemm_trace.txt
SA_LEAN=0 bash scripts/gemm_trace.sh 0 > $O/gemm_trace_notlean.txt 2>&1; cat $O/gemm_trace_notlean.txt
WORKLOAD=c2e bash scripts/gemm_trace.sh 0 > $O/gemm_trace_c2e.txt 2>&1; cat $O/gemm_trace_c2e.txt
echo DONE
