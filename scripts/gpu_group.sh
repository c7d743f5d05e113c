#!/bin/bash
# the device group against the single-engine tracker on one GPU, and the host-manners knobs
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; : > gpurun_out/group_tracker.jsonl
for dv in - 0,0 0,0,0,0; do
  for cfg in "sort 8 500" "sort 64 500"; do
    for md in sync async; do
      python scripts/bench_batch_tracker.py $cfg 256 60 0 $md rows $dv 2>/dev/null | tail -1 >> gpurun_out/group_tracker.jsonl
    done
  done
  python scripts/bench_batch_tracker.py visual 8 1000 512 30 0 sync rows $dv 2>/dev/null | tail -1 >> gpurun_out/group_tracker.jsonl
done
python scripts/bench_batch_tracker.py sort 8 500 256 60 -4 sync rows - 0 2>/dev/null | tail -1 >> gpurun_out/group_tracker.jsonl
python scripts/bench_batch_tracker.py sort 64 500 256 60 -4 async rows - 0 2>/dev/null | tail -1 >> gpurun_out/group_tracker.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/group_tracker.jsonl"):
    d=json.loads(l)
    print(d["tracker"], d["scenes"], "x", d["objects_per_scene"], d["call"], "devices", d["devices"], "workers", d["workers"], "spin", d["spin_us"], "->", d["us_per_predict_median"], "us  cpu-s/1000", d["cpu_seconds_per_1000_predicts"], "cores", d["cpu_cores_busy_over_the_loop"], "begin", d.get("us_until_begin_returns"), "first", d.get("us_until_first_scene"))
PY
