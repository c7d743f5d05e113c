#!/bin/bash
# Device-side timeline of the facade loop: rocprofv3 kernel trace of scripts/bench_tracker.py (one configuration), then per predict()
# the kernels in order with their durations and the gaps between them (medians over the frames).
#   scripts/tracker_timeline.sh "visual,device,0.0"
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
export TMPDIR=/tmp
cfg="$1"; tag=$(echo "$cfg" | tr ',.' '__')
rm -rf gpurun_out/tl_$tag
SA_BENCH_TRACKER_ONLY="$cfg" timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$tag -o t -- python scripts/bench_tracker.py 1000 512 40 > gpurun_out/tl_$tag.out 2>&1
f=$(find gpurun_out/tl_$tag -name '*kernel_trace.csv' | head -1)
python - "$f" "$cfg" <<'PY'
import csv, sys, collections, numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"].split("(")[0][:60], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# a predict() = the kernels between two gaps of more than 15 us ... group by the first-phase kernel instead
first = [i for i, e in enumerate(ev) if "k_frame" in e[0]]
frames = [ev[a:b] for a, b in zip(first[:-1], first[1:])][-25:]
sig = collections.Counter(tuple(k[0] for k in fr) for fr in frames).most_common(1)[0][0]
frames = [fr for fr in frames if tuple(k[0] for k in fr) == sig]
print(sys.argv[2], "frames with the common kernel sequence:", len(frames))
for j, name in enumerate(sig):
    dur = np.median([(fr[j][2] - fr[j][1]) / 1e3 for fr in frames])
    gap = np.median([(fr[j][1] - fr[j - 1][2]) / 1e3 for fr in frames]) if j else float("nan")
    print(f"   {name:60s} {dur:7.2f} us   gap before {gap:7.2f}")
per = np.median([(b[0][1] - a[0][1]) / 1e3 for a, b in zip(frames[:-1], frames[1:])])
busy = np.median([sum(k[2] - k[1] for k in fr) / 1e3 for fr in frames])
span = np.median([(fr[-1][2] - fr[0][1]) / 1e3 for fr in frames])
print(f"   first launch to first launch {per:.1f} us | kernels busy {busy:.1f} | first begin -> last end {span:.1f}")
PY
