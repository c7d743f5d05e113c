#!/bin/bash
# One GPU-box visit that produces everything judged under profiles/: bench lines for C2..C5, rocprofv3 kernel stats of the
# default bench command, MFMA-utilisation PMC pass of the contraction, HBM-traffic PMC passes.   scripts/profile_round.sh <tag>
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r01_e}; O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
(rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6; nproc) > $O/env.txt 2>&1
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
for w in c1 c2e c2k3 c3 c3m c4 c5 sd; do python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; done
python bench.py --flags 32 --no-cpu-baseline > $O/bench_c2_separate.json 2> $O/bench_c2_separate.err
for w in c2 c2k3 c5; do python bench.py --workload $w --flags 64 --no-cpu-baseline > $O/bench_${w}_f16split.json 2> $O/bench_${w}_f16split.err; done
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_mfma_c2sep -o p -- python $OLDPWD/bench.py --workload c2 --flags 32 --steps 20 --warmup 3 --no-cpu-baseline --profile-iters 5 > $O/pmc_mfma_c2sep.log 2>&1)
for w in c2 c5; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o bench -- python $OLDPWD/bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline > $O/prof_$w.log 2>&1)
  f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${w}_kernel_stats.csv
  find $O/prof_$w -name "*kernel_trace.csv" -size +4M -delete
  (cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_mfma_$w -o p -- python $OLDPWD/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --profile-iters 5 > $O/pmc_mfma_$w.log 2>&1)
done
python - "$O" <<'PY'
import csv, glob, sys, collections, json
o = sys.argv[1]
out = {}
for w in ("c2", "c2sep", "c5"):
    for f in glob.glob(f"{o}/pmc_mfma_{w}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in agg.items():
            if "visual_cos" not in k and "k_frame_visual" not in k: continue
            m = {c: sum(v) / len(v) for c, v in d.items()}
            # MFMA busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs
            util = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (m["GRBM_GUI_ACTIVE"] / 8.0)
            out[w] = {"kernel": k, "counters": m, "mfma_pipe_busy_fraction": util}
json.dump(out, open(f"{o}/mfma_util.json", "w"), indent=1)
print(json.dumps({w: round(v["mfma_pipe_busy_fraction"], 3) for w, v in out.items()}))
PY
bash scripts/pmc_traffic.sh c2 ${TAG}_c2 > /dev/null 2>&1; bash scripts/pmc_traffic.sh c4 ${TAG}_c4 > /dev/null 2>&1; bash scripts/pmc_traffic.sh c5 ${TAG}_c5 > /dev/null 2>&1
python scripts/bench_tracker.py 1000 512 30 > $O/tracker_loop.jsonl 2>/dev/null; python scripts/bench_tracker.py 500 128 30 >> $O/tracker_loop.jsonl 2>/dev/null
python scripts/bench_nms.py > $O/nms.jsonl 2>/dev/null; python scripts/bench_own_areas.py > $O/own_areas.jsonl 2>/dev/null
cat $O/bench_c2.json | cut -c1-600
