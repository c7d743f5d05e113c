#!/bin/bash
# Where a facade predict() spends its time (SA_TRACKER_TRACE=1: one line per call on stderr), medians over the frames of one run.
#   scripts/tracker_trace.sh "visual,device,0.0" "sort,rows,0.0" ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
for cfg in "$@"; do
  SA_TRACKER_TRACE=1 SA_BENCH_TRACKER_ONLY="$cfg" timeout 300 python scripts/bench_tracker.py 1000 512 30 > gpurun_out/tt.out 2> gpurun_out/tt.err
  python - "$cfg" <<'PY'
import re, sys, numpy as np
rows = []
for ln in open("gpurun_out/tt.err"):
    if ln.startswith("[sa_tracker]"):
        rows.append([float(x) for x in re.findall(r"(-?[0-9]+\.[0-9]+)", ln)])
a = np.array(rows[5:])
names = ["assemble", "associate", "begin", "stage", "enqueue", "wait+fetch", "apply", "bookkeeping"]
print(sys.argv[1], open("gpurun_out/tt.out").read().strip())
col = np.array([[float(x) for x in re.findall(r"(-?[0-9]+[.][0-9]+)", ln)] for ln in open("gpurun_out/tt.err") if ln.startswith("[sa_collect]")][5:])
if len(col): print("   collect:", {n: round(float(np.median(col[:, i])), 1) for i, n in enumerate(("sync", "table", "polygons"))})
print("   median us per predict():", {n: round(float(np.median(a[:, i])), 1) for i, n in enumerate(names)}, "sum", round(float(np.median(a[:, 0] + a[:, 1] + a[:, 6] + a[:, 7])), 1))
PY
done
