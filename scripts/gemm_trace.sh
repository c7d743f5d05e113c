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
# s_memtime has its own base on every XCD and workgroups go round-robin over the 8 XCDs: times are compared inside an XCD only
xcd=a[:,0].astype(int)%8
rel=a.copy()
for x in range(8):
    m=xcd==x
    if m.any(): rel[m,1:]=np.where(a[m,1:]>0, a[m,1:]-a[m,1][a[m,1]>0].min(), 0)
isother=(a[:,2]==1)|(a[:,2]==2)
if isother.any():
    for kind,name in ((2,"preparation blocks"),(1,"positional tiles")):
        o=rel[a[:,2]==kind]
        if len(o): print("   %s: %d blocks, entry %.0f .. %.0f (median %.0f), exit %.0f .. %.0f, mean life %.0f cycles" % (name,len(o),o[:,1].min(),o[:,1].max(),np.median(o[:,1]),o[:,6].min(),o[:,6].max(),(o[:,6]-o[:,1]).mean()))
    g=rel[~isother]
    print("   contraction tiles: entry %.0f .. %.0f, exit %.0f .. %.0f" % (g[:,1].min(),g[:,1].max(),g[:,6].min(),g[:,6].max()))
    pct=lambda v: " / ".join("%.0f" % np.percentile(v,q) for q in (10,50,90,99,100))
    o=rel[a[:,2]==1]
    if len(o): print("   exit percentiles 10/50/90/99/100: positional tiles", pct(o[:,6]), "| contraction tiles", pct(g[:,6]), "| positional life", pct(o[:,6]-o[:,1]))
a=a[~isother]
t=a[:,1:7]
d=np.diff(t,axis=1)
print("flags $fl", "tiles",len(a),"prologue/main/reduce/epilogue/maxkey:",d.mean(0).round(0), "total",(t[:,5]-t[:,0]).mean().round(0))
if a.shape[1] > 8 and a[:,7].min() > 0:
    print("   epilogue split: operands + cells", (a[:,7]-a[:,4]).mean().round(0), "rows -> partials", (a[:,8]-a[:,7]).mean().round(0), "column partials", (a[:,5]-a[:,8]).mean().round(0))
PY
done
