cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for mode in tile part; do
  rm -f gpurun_out/gemm_trace.txt
  if [ $mode = tile ]; then export SA_BESTFIT=tile; else unset SA_BESTFIT; fi
  SA_GEMM_TRACE=30 timeout 300 python bench.py --workload c2 --no-cpu-baseline --steps 40 --warmup 5 > /dev/null 2>&1
  python - <<PY
import numpy as np
a=np.loadtxt("gpurun_out/gemm_trace.txt")
t=a[:,1:7]
d=np.diff(t,axis=1)
print("$mode", "blocks",len(a),"phases prologue/main/reduce/epilogue/maxkey:",d.mean(0).round(0), "total",(t[:,5]-t[:,0]).mean().round(0))
if a.shape[1] > 8 and a[:,7].min() > 0:
    print("   epilogue split: cells+reductions", (a[:,7]-a[:,4]).mean().round(0), "barrier", (a[:,8]-a[:,7]).mean().round(0), "partial stores", (a[:,5]-a[:,8]).mean().round(0))
PY
done
