#!/bin/bash
# In-kernel timeline of the contraction's tiles (library built with SA_EXTRA_FLAGS=-DSA_GEMM_TRACE): per-phase cycles, averaged
# over the tiles of one launch, for the engine flag sets given as arguments (default: 0 = heterogeneous first phase, 32 = separate)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
for fl in ${@:-0 32}; do
  rm -f gpurun_out/gemm_trace.txt
  SA_GEMM_TRACE=30 timeout 300 python bench.py --workload ${WORKLOAD:-c2} --no-cpu-baseline --steps 40 --warmup 5 --flags $fl > /dev/null 2>&1
  python - <<PY
import numpy as np
a=np.loadtxt("gpurun_out/gemm_trace.txt")
a=a[a[:,1] > a[:,1].max() - 1e6]   # stamps of the traced launch only (the buffer also holds blocks of earlier, larger launches)
other=a[(a[:,2]==1)|(a[:,2]==2)]
a=a[(a[:,2]!=1)&(a[:,2]!=2)]
if len(other):
    t0=min(a[:,1].min(), other[:,1].min())
    for kind,name in ((1,"positional tiles"),(2,"preparation blocks")):
        o=other[other[:,2]==kind]
        if len(o): print("   %s: %d blocks, entry %.0f..%.0f, exit %.0f..%.0f (mean life %.0f) cycles after the first entry" % (name,len(o),(o[:,1]-t0).min(),(o[:,1]-t0).max(),(o[:,6]-t0).min(),(o[:,6]-t0).max(),(o[:,6]-o[:,1]).mean()))
    print("   contraction tiles: entry %.0f..%.0f, exit %.0f..%.0f" % ((a[:,1]-t0).min(),(a[:,1]-t0).max(),(a[:,6]-t0).min(),(a[:,6]-t0).max()))
t=a[:,1:7]
d=np.diff(t,axis=1)
print("flags $fl", "tiles",len(a),"prologue/main/reduce/epilogue/maxkey:",d.mean(0).round(0), "total",(t[:,5]-t[:,0]).mean().round(0), "first entry -> last exit", t[:,5].max()-t[:,0].min(), "entry spread", t[:,0].max()-t[:,0].min())
if a.shape[1] > 8 and a[:,7].min() > 0:
    print("   epilogue split: operands + cells", (a[:,7]-a[:,4]).mean().round(0), "rows -> partials", (a[:,8]-a[:,7]).mean().round(0), "column partials", (a[:,5]-a[:,8]).mean().round(0))
PY
done
