#!/bin/bash
# scripts/idle_gap_probe.py under a kernel trace: average kernel durations per phase (phases are separated by > 10 ms without kernels).
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for w in ${1:-c3 c2}; do
  rm -rf gpurun_out/idle_gap_$w
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/idle_gap_$w -o t -- python scripts/idle_gap_probe.py $w ${2:-60} > gpurun_out/idle_gap_$w.out 2>&1
  tail -n 1 gpurun_out/idle_gap_$w.out
  f=$(find gpurun_out/idle_gap_$w -name '*kernel_trace.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
phases, cur, prev_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None and s - prev_end > 10_000_000:
        phases.append(cur); cur = []
    cur.append((r["Kernel_Name"].split("(")[0][:40], (e - s) / 1e3))
    prev_end = e
phases.append(cur)
for i, ph in enumerate(phases[-3:]):
    agg = collections.defaultdict(list)
    for k, d in ph: agg[k].append(d)
    print("   phase", "ABC"[i], {k: (round(sum(v) / len(v), 2), len(v)) for k, v in agg.items()})
PY
done
