#!/bin/bash
# Per-tile timeline of the positional launch k_frame (library built with SA_EXTRA_FLAGS=-DSA_POS_TRACE): phase durations in cycles
# (s_memtime, 100 MHz-independent shader clock) over the tiles of ONE launch, percentiles -> gpurun_out/pos_trace_<workload>.txt
#   WORKLOADS="c4 c3" scripts/pos_trace.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
for w in ${WORKLOADS:-c4 c3}; do
  rm -f gpurun_out/pos_trace.txt
  SA_POS_TRACE=30 timeout 300 python bench.py --workload $w --no-cpu-baseline --no-oracle --no-h2d --steps 40 --warmup 5 > /dev/null 2> gpurun_out/pt_$w.err
  python - "$w" <<'PY' | tee gpurun_out/pos_trace_$w.txt
import sys, numpy as np
w = sys.argv[1]
a = np.loadtxt("gpurun_out/pos_trace.txt")
a = a[a[:, 5] > 0]   # tiles that ran to the end
blk, t = a[:, 0].astype(int), a[:, 1:6]
d = np.diff(t, axis=1)
pct = lambda v: " / ".join("%6.0f" % np.percentile(v, q) for q in (10, 50, 90, 99, 100))
print(f"== {w}: {len(a)} positional tiles of one k_frame launch; cycles (s_memtime), percentiles 10 / 50 / 90 / 99 / max")
for i, name in enumerate(("boxes -> LDS + track loads", "screen (too_far, compatible) + survivor list", "disjointness proofs", "clip rounds + edge append")):
    print(f"   {name:46s} {pct(d[:, i])}")
print(f"   {'tile life (entry -> last edge out)':46s} {pct(t[:, 4] - t[:, 0])}")
print(f"   survivors per tile {pct(a[:, 7])} | pairs clipped per tile {pct(a[:, 8])}")
# occupancy timeline inside each XCD (s_memtime bases differ between XCDs; workgroups go round-robin over the 8 XCDs)
xcd = blk % 8
for x in range(8):
    m = xcd == x
    if not m.any(): continue
    base = t[m, 0].min()
    ent, ext = t[m, 0] - base, t[m, 4] - base
    print(f"   XCD {x}: {m.sum():5d} tiles, entries {pct(ent)} | exits {pct(ext)}")
PY
done
