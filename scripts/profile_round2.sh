#!/bin/bash
# One GPU-box visit that produces everything judged under profiles/ for a round-2+ state of the repo: bench lines of every workload
# (each with value_resident / value_h2d / match_vs_oracle), rocprofv3 kernel stats of the default bench command and of C2e / C5,
# MFMA-utilisation PMC passes, HBM-traffic PMC passes, the two-rank dispatch rehearsal, the in-process cluster, the side benches.
#   scripts/profile_round2.sh <tag>     -> gpurun_out/<tag>/ ; copy what is to be judged into profiles/<tag>_*
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r02_z}; O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
(rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6; nproc) > $O/env.txt 2>&1
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver_command.json 2> $O/bench_c2_driver_command.err   # the driver's own command line
for w in c1 c2n c2e c2k3 c2d c3 c3m c4 sd; do timeout 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; done
for w in giant bigcrowd bigpile; do timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 3 --profile-iters 10 > $O/bench_$w.json 2> $O/bench_$w.err; done   # milliseconds per step: short regions
timeout 600 python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
timeout 300 python bench.py --flags 32 --no-cpu-baseline --no-oracle > $O/bench_c2_separate.json 2> $O/bench_c2_separate.err
SA_EUCLID=valu timeout 300 python bench.py --workload c2e --no-cpu-baseline --no-oracle > $O/bench_c2e_valu.json 2> $O/bench_c2e_valu.err
timeout 300 python bench.py --workload c3 --no-cpu-baseline --no-oracle --cluster 2 --cluster-devices 0,0 > $O/bench_c3_cluster2.json 2> $O/bench_c3_cluster2.err
# two ranks on the one GPU of this box (gloo collectives): the N > 1 control flow, dispatch pass included
SA_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload c3 --steps 50 --warmup 5 --no-cpu-baseline --no-oracle > $O/bench_c3_2ranks_one_device.json 2> $O/bench_c3_2ranks_one_device.err
for w in c2 c2e c5; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o bench -- python $OLDPWD/bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --no-oracle > $O/prof_$w.log 2>&1)
  f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${w}_kernel_stats.csv
  find $O/prof_$w -name "*kernel_trace.csv" -size +4M -delete
  (cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_mfma_$w -o p -- python $OLDPWD/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-oracle --no-h2d --profile-iters 5 > $O/pmc_mfma_$w.log 2>&1)
done
python - "$O" <<'PY'
import csv, glob, sys, collections, json
o = sys.argv[1]
out = {}
for w in ("c2", "c2e", "c5"):
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
for w in c2 c2e c4 c5 sd; do bash scripts/pmc_traffic.sh $w ${TAG}_$w > $O/pmc_traffic_$w.log 2>&1; done
python scripts/bench_tracker.py 1000 512 30 > $O/tracker_loop.jsonl 2>/dev/null; python scripts/bench_tracker.py 500 128 30 >> $O/tracker_loop.jsonl 2>/dev/null
python scripts/bench_nms.py > $O/nms.jsonl 2>/dev/null; python scripts/bench_own_areas.py > $O/own_areas.jsonl 2>/dev/null
cut -c1-400 $O/bench_c2.json; tail -2 $O/bench_c3_2ranks_one_device.err; cut -c1-300 $O/bench_c3_2ranks_one_device.json
