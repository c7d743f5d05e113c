#!/bin/bash
# Where a facade predict() spends its time (SA_TRACKER_TRACE=1: the facade prints its phases per call on stderr — two lines for the fused
# path: up to the launches / behind them), medians over the frames of one run.
#   scripts/tracker_trace.sh "visual,device,0.0" "sort,rows,0.0" ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
for cfg in "$@"; do
  SA_TRACKER_TRACE=1 SA_BENCH_TRACKER_ONLY="$cfg" timeout 300 python scripts/bench_tracker.py 1000 512 30 > gpurun_out/tt.out 2> gpurun_out/tt.err
  python - "$cfg" <<'PY'
import re, sys, statistics as st
L = [l for l in open("gpurun_out/tt.err") if l.startswith("[sa_tracker]")]
print(sys.argv[1], open("gpurun_out/tt.out").read().strip())
names = {"up to the launches": ["before the jobs", "assemble", "longest assemble job", "epochs", "stage + evict", "enqueue"],
         "behind the launches": ["deferred", "wait for the association", "merges", "longest merge job", "all merge jobs", "wait for the Kalman dispatch",
                                 "tables + results", "longest job", "all jobs", "minor faults"]}
total = 0.0
for key, nm in names.items():
    rows = [[float(x) for x in re.findall(r"[-+]?\d+\.\d+", l)] for l in L if key in l][5:]
    rows = [r for r in rows if len(r) == len(nm)]
    if rows:
        med = dict(zip(nm, [round(st.median(c), 1) for c in zip(*rows)]))
        print("   ", key + " (median us):", med)
        total += sum(v for k, v in med.items() if k in ("before the jobs", "assemble", "epochs", "stage + evict", "enqueue", "deferred", "wait for the association", "merges",
                                                         "wait for the Kalman dispatch", "tables + results"))
other = [l for l in L if "up to the launches" not in l and "behind the launches" not in l]
if other:
    print("    (%d lines of the general path; last: %s)" % (len(other), other[-1].strip()[:200]))
if total:
    print("    sum of the phases: %.1f us" % total)
PY
done
