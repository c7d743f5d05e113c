#!/bin/bash
# HBM traffic per kernel launch of the bench pipeline, from the L2 memory-side counters.  Separate --pmc passes with
# --kernel-trace only (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: they do not fit one pass; MI355X_MICROARCH.md §HBM,
# §rocprofv3 PMC slots).  Usage: scripts/pmc_traffic.sh <workload> <tag>   -> gpurun_out/pmc_<tag>_{fetch,write}.csv + summary json
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
W=${1:-c2}; TAG=${2:-r01}
OUT=$PWD/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o p -- \
     python $OLDPWD/bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-oracle --no-h2d --profile-iters 5 ${SA_BENCH_FLAGS:+--flags $SA_BENCH_FLAGS} > $OUT/$C.log 2>&1)
done
python - "$OUT" "$W" <<'PY'
import csv, glob, json, sys, collections
out, w = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            res[k][c] = {"launches": len(v), "avg": sum(v) / len(v)}
json.dump({"workload": w, "unit": "KiB per launch as reported by rocprofv3 (FETCH_SIZE must be doubled for wide coalesced reads on gfx950)",
           "kernels": res}, open(f"{out}/summary.json", "w"), indent=1)
for k, d in sorted(res.items()):
    print(k[:60], {c: round(x["avg"], 1) for c, x in d.items()})
PY
