#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_h; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "cosine_parity or full_size_c2 or vote_words or euclid" > $O/pytest_first.log 2>&1; echo "first exit $?" >> $O/pytest_first.log; tail -6 $O/pytest_first.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h = d.get("h2d_inclusive") or {}
    print(sys.argv[2], {k: d.get(k) for k in ("ms_per_step", "value_h2d", "match_vs_oracle")}, {k: (round(v["avg_us"], 2), round(v["avg_us_instrumented"], 2)) for k, v in d.get("kernels", {}).items()}, (d.get("roofline") or {}).get("frac"), h.get("ms_per_step"))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for m in side fused serial; do
  if [ $m = side ]; then unset SA_FIRST_PHASE; else export SA_FIRST_PHASE=$m; fi
  for w in c2 c2e c2n c2k3; do timeout 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_${w}_$m.json 2> $O/bench_${w}_$m.err; show $O/bench_${w}_$m.json "$w first_phase=$m"; done
done
unset SA_FIRST_PHASE
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o bench -- python $OLDPWD/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-oracle --no-h2d > $O/prof_c2.log 2>&1)
f=$(find $O/prof_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c2_kernel_stats.csv && head -5 $O/c2_kernel_stats.csv | cut -c1-160
python - $O <<'PY'
import csv, glob, sys
o = sys.argv[1]
f = glob.glob(f"{o}/prof_c2/**/*kernel_trace.csv", recursive=True)
if f:
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:34]) for r in csv.DictReader(open(f[0]))]
    rows.sort()
    a = len(rows) // 2
    t0 = rows[a][0]
    for r in rows[a:a + 12]:
        print(f"{(r[0]-t0)/1e3:9.1f} {(r[1]-t0)/1e3:9.1f} {(r[1]-r[0])/1e3:7.1f} {r[2]}")
PY
rm -rf $O/prof_c2
echo DONE
