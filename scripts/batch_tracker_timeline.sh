#!/bin/bash
# Device-side timeline of Batch*::predict through the facade: rocprofv3 kernel trace of scripts/bench_batch_tracker.py, then per predict()
# the kernels in order with their durations and the gaps between them (medians over the calls), and the span first begin -> last end —
# the device time a predict() of the facade is measured against.
#   scripts/batch_tracker_timeline.sh <tag> sort 8 500 256 30 0 sync
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=$1; shift
rm -rf gpurun_out/btl_$tag
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/btl_$tag -o t -- python scripts/bench_batch_tracker.py "$@" > gpurun_out/btl_$tag.out 2>&1
tail -n 1 gpurun_out/btl_$tag.out
f=$(find gpurun_out/btl_$tag -name '*kernel_trace.csv' | head -1)
python - "$f" "$*" <<'PY'
import csv, sys, collections, numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"].split("(")[0][:60], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
first = [i for i, e in enumerate(ev) if e[0].startswith("k_ingest") or e[0].startswith("void k_ingest")]
if len(first) < 8:
    first = [i for i, e in enumerate(ev) if "k_frame" in e[0]]
frames = [ev[a:b] for a, b in zip(first[:-1], first[1:])][-24:-1]
sig = collections.Counter(tuple(k[0] for k in fr) for fr in frames).most_common(1)[0][0]
frames = [fr for fr in frames if tuple(k[0] for k in fr) == sig]
print(sys.argv[2], "| calls with the common kernel sequence:", len(frames))
for j, name in enumerate(sig):
    dur = np.median([(fr[j][2] - fr[j][1]) / 1e3 for fr in frames])
    gap = np.median([(fr[j][1] - fr[j - 1][2]) / 1e3 for fr in frames]) if j else float("nan")
    print(f"   {name:60s} {dur:8.2f} us   gap before {gap:7.2f}")
per = np.median([(b[0][1] - a[0][1]) / 1e3 for a, b in zip(frames[:-1], frames[1:])]) if len(frames) > 1 else float("nan")
busy = np.median([sum(k[2] - k[1] for k in fr) / 1e3 for fr in frames])
span = np.median([(fr[-1][2] - fr[0][1]) / 1e3 for fr in frames])
print(f"   first launch to first launch {per:.1f} us | kernels busy {busy:.1f} | first begin -> last end (device time of a predict) {span:.1f}")
PY
