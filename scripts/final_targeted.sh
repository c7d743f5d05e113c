cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=$PWD/gpurun_out/r06_zzz; rm -rf $O; mkdir -p $O
for w in sdt c4 c2t; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o bench -- python $OLDPWD/bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --no-oracle --no-h2d > $O/prof_$w.log 2>&1)
  f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${w}_kernel_stats.csv; rm -rf $O/prof_$w
  timeout 600 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
done
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver_command.json 2> $O/bench_c2_driver_command.err
timeout 600 python scripts/bench_tracker.py 1000 512 30 > $O/tracker_loop.jsonl 2> $O/tracker_loop.err
python - $O <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1]); r = d.get("roofline") or {}
    print(os.path.basename(f), round(d["ms_per_step"]*1e3, 2), "frac", r.get("frac"), "rocprof", r.get("frac_rocprof"), {k: round(x["avg_us"], 2) for k, x in d.get("kernels", {}).items()}, "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
cat $O/tracker_loop.jsonl | cut -c1-250
