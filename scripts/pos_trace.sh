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
a = np.loadtxt("gpurun_out/pos_trace.txt", dtype=np.uint64)
a = a[a[:, 5] > 0]   # tiles that ran to the end (column 5 = stamp 4)
blk, t = a[:, 0].astype(int), a[:, 1:6].astype(np.int64)
d = np.diff(t, axis=1)
pct = lambda v: " / ".join("%6.0f" % np.percentile(v, q) for q in (10, 50, 90, 99, 100))
print(f"== {w}: {len(a)} positional tiles of one k_frame launch; cycles (s_memtime), percentiles 10 / 50 / 90 / 99 / max")
for i, name in enumerate(("boxes -> LDS + track loads", "screen (too_far, compatible) + survivor list", "disjointness proofs", "clip rounds + edge append")):
    print(f"   {name:46s} {pct(d[:, i])}")
print(f"   {'tile life (entry -> last edge out)':46s} {pct(t[:, 4] - t[:, 0])}")
print(f"   survivors per tile {pct(a[:, 7].astype(np.int64))} | pairs clipped per tile {pct(a[:, 8].astype(np.int64))}")
# when the tiles ran, on the 100 MHz clock all XCDs share ([5] = entry << 32 | exit): microseconds from the first tile's entry
rt = a[:, 6]
ent, ext = (rt >> np.uint64(32)).astype(np.int64), (rt & np.uint64(0xffffffff)).astype(np.int64)
ent = ent & 0xffffffff
ext = np.where(ext < ent, ext + (1 << 32), ext)
t0 = ent.min()
print(f"   tile entry, us after the first tile's   {' / '.join('%6.2f' % (np.percentile(ent - t0, q) / 100.0) for q in (10, 50, 90, 99, 100))}")
print(f"   tile exit,  us after the first tile's   {' / '.join('%6.2f' % (np.percentile(ext - t0, q) / 100.0) for q in (10, 50, 90, 99, 100))}")
PY
done
