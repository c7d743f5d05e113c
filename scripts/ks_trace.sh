#!/bin/bash
# In-kernel timeline of the stand-alone contraction (k_cosine_matrix) per tile plan: rebuilds the library with -DSA_GEMM_TRACE on the box.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
SA_EXTRA_FLAGS=-DSA_GEMM_TRACE python -m similari_amd.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
for plan in ${@:-1 10 13}; do
  for shape in ${SHAPES:-c2}; do
  rm -f gpurun_out/gemm_trace.txt
  SA_GEMM_TRACE=20 timeout 300 python scripts/gemm_bench.py $shape $plan > /dev/null 2>&1
  python - <<PY
import numpy as np
a=np.loadtxt("gpurun_out/gemm_trace.txt")
t=a[:,1:6]
d=np.diff(t,axis=1)
print("plan $plan $shape tiles",len(a),"prologue/main/reduce/epilogue:",d.mean(0).round(0),"total",(t[:,4]-t[:,0]).mean().round(0), "main p10/p50/p90", np.percentile(d[:,1],[10,50,90]).round(0))
PY
  done
done
