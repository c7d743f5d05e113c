#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_f; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cluster.py -m gpu -q -p no:cacheprovider > $O/pytest_pipe.log 2>&1; echo "exit $?" >> $O/pytest_pipe.log; tail -5 $O/pytest_pipe.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h = d.get("h2d_inclusive") or {}
    print(sys.argv[2], {k: d.get(k) for k in ("ms_per_step", "value_h2d")}, {k: h.get(k) for k in ("ms_per_step", "synchronous_ms_per_step", "host_us_in_submit", "host_us_in_wait", "tickets_in_flight")})
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for d in 3 2; do for ing in kernel sdma; do
  if [ $ing = sdma ]; then export SA_INGEST=sdma; else unset SA_INGEST; fi
  SA_PIPE_DEPTH=$d timeout 300 python bench.py --no-cpu-baseline --no-oracle > $O/bench_c2_d${d}_$ing.json 2> $O/bench_c2_d${d}_$ing.err; show $O/bench_c2_d${d}_$ing.json "c2 depth=$d ingest=$ing"
done; done
unset SA_INGEST
for w in c3 c5 c2e; do timeout 600 python bench.py --workload $w --no-cpu-baseline --no-oracle > $O/bench_$w.json 2> $O/bench_$w.err; show $O/bench_$w.json "$w"; done
# timeline of the pipelined loop: kernel trace of a short run (c2, 60 steps)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $OLDPWD/bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-oracle --profile-iters 2 > $O/trace.log 2>&1)
python - $O <<'PY'
import csv, glob, sys
o = sys.argv[1]
f = glob.glob(f"{o}/trace/**/*kernel_trace.csv", recursive=True)
if f:
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:28]) for r in csv.DictReader(open(f[0]))]
    rows.sort()
    ing = [i for i, r in enumerate(rows) if "k_ingest" in r[2]]
    # the pipelined phase: the longest run of ingests with a first-phase kernel between each pair
    if len(ing) > 40:
        a, b = ing[len(ing) // 2 - 8], ing[len(ing) // 2 + 8]
        t0 = rows[a][0]
        with open(f"{o}/timeline.txt", "w") as w:
            for r in rows[a:b + 1]:
                w.write(f"{(r[0]-t0)/1e3:9.1f} {(r[1]-t0)/1e3:9.1f} {(r[1]-r[0])/1e3:7.1f} {r[2]}\n")
        print(open(f"{o}/timeline.txt").read())
PY
rm -rf $O/trace
echo DONE
