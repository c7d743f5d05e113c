#!/bin/bash
# The whole GPU suite, smoke, the default bench line, then the facade's lines of the profile set (scripts/profile_round.sh section 5 / 5b).
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r05_refresh}; O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=5 -rf > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | cut -c1-250 | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench exit $?"
timeout 600 python scripts/bench_tracker.py 1000 512 30 > $O/tracker_loop.jsonl 2> $O/tracker_loop.err
J=$O/batch_tracker.jsonl; : > $J
brun() { timeout 300 python scripts/bench_batch_tracker.py "$@" >> $J 2>> $O/batch_err.txt || echo "bench_batch_tracker $* failed"; }
brun sort 8 500 0 60 0 sync; brun sort 8 500 0 60 1 sync; brun sort 8 500 0 60 0 async
brun sort 64 500 0 60 0 sync; brun sort 64 500 0 60 1 sync; brun sort 64 500 0 60 8 sync; brun sort 64 500 0 60 0 async
brun visual 8 1000 512 24 0 sync device; brun visual 8 1000 512 24 1 sync device; brun visual 8 1000 512 24 0 sync rows; brun visual 8 1000 512 24 0 async device
python - $J <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print({k: d[k] for k in ("tracker", "scenes", "workers", "call", "features", "us_per_predict_median", "us_until_begin_returns", "us_until_first_scene") if k in d})
PY
g++ -O2 -std=c++17 -I include scripts/micro/batch_handle_bench.cpp -L similari_amd/lib -lsimilari_assoc -Wl,-rpath,$PWD/similari_amd/lib -o /tmp/bhb && { /tmp/bhb 8 500 80; /tmp/bhb 64 500 60; } > $O/batch_handle_cpp.jsonl; cat $O/batch_handle_cpp.jsonl
python - $O/bench_c2.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "value_h2d")}, "device_features", d.get("device_features_inclusive", {}).get("pairs_per_s"), "roofline", {k: d["roofline"].get(k) for k in ("frac", "frac_rocprof", "traffic")}, "match", d.get("match_vs_oracle"))
PY
