#!/bin/bash
# What do the kernels of sa_batch_time's replay look like behind a 64-scene BatchSort loop? (kernel trace, last 40 dispatches)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf gpurun_out/replay_probe
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/replay_probe -o t -- python scripts/bench_batch_tracker.py sort 64 500 0 12 0 sync > gpurun_out/replay_probe.out 2>&1
tail -n 1 gpurun_out/replay_probe.out
f=$(find gpurun_out/replay_probe -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows[-48:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{r["Kernel_Name"].split("(")[0][:50]:50s} {(e - s) / 1e3:8.2f} us  gap {((s - prev) / 1e3) if prev else 0:8.2f}  grid {r.get("Grid_Size_X")}x{r.get("Grid_Size_Y")}x{r.get("Grid_Size_Z")} wg {r.get("Workgroup_Size_X")}')
    prev = e
PY
