#!/bin/bash
# tracker facade timing with its per-frame breakdown
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r03_t}; O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_trackers.py tests/test_gpu_pipeline.py tests/test_gpu_cluster.py -m gpu -q -p no:cacheprovider --maxfail=10 -rf > $O/pytest_trk.log 2>&1; echo "trk exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_trk.log | cut -c1-300 | tail -12
SA_TRACKER_TRACE=1 timeout 300 python scripts/bench_tracker.py 1000 512 30 > $O/tracker_loop.jsonl 2> $O/tracker_trace.txt; echo "tracker exit $?"
cat $O/tracker_loop.jsonl
SA_SYNC=block timeout 300 python scripts/bench_tracker.py 1000 512 30 > $O/tracker_loop_block.jsonl 2>/dev/null; echo "--- SA_SYNC=block"; cat $O/tracker_loop_block.jsonl
# median of each trace field per run (4 runs x 30 frames in order: sort host, sort device, visual host, visual device)
python - $O/tracker_trace.txt <<'PY'
import re, sys, statistics as st
lines=[l for l in open(sys.argv[1]) if l.startswith("[sa_tracker]")]
names=["sort host","sort device","visual host rows","visual device rows","visual device pinned","visual device devblock"]
n=len(lines)//len(names)
for r in range(len(names)):
    chunk=lines[r*n+5:(r+1)*n]
    vals=[[float(x) for x in re.findall(r"[-+]?\d+\.\d+", l)] for l in chunk]
    med=[round(st.median(c),1) for c in zip(*vals)]
    print(names[r], dict(zip(["assemble","associate","begin","stage","enqueue","wait+fetch","apply","bookkeeping"], med)))
PY
shift
for w in "$@"; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w exit $?"
  python - "$O/bench_$w.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "match_vs_oracle")}, {k: round(v["avg_us"], 2) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print("unreadable", e)
PY
done
echo DONE
