#!/bin/bash
# Vector-pipe issue demand of the first phase's kernels (the f32 matrix instruction and the vector instructions share a SIMD's issue:
# DESIGN §3): SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU / SQ_VALU_MFMA_BUSY_CYCLES per kernel, fused launch and the two kernels apart.
# Usage: scripts/pmc_valu.sh [workload]   -> gpurun_out/pmc_valu.txt
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; W=${1:-c2}; O=$PWD/gpurun_out/pmc_valu; rm -rf $O; mkdir -p $O
for mode in fused apart; do
  fl=""; [ $mode = apart ] && fl="--flags 32"
  for set in "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
    tag=$(echo $set | cut -d' ' -f1)
    (cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${mode}_$tag -o p -- \
       python $OLDPWD/bench.py --workload $W $fl --steps 20 --warmup 5 --no-cpu-baseline --no-oracle --no-h2d > $O/${mode}_$tag.log 2>&1)
  done
done
python - $O <<'PY' | tee gpurun_out/pmc_valu.txt
import csv, glob, sys, collections
O = sys.argv[1]
for mode in ("fused", "apart"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{mode}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(acc.items()):
        if not any(s in k for s in ("k_frame", "k_visual", "k_assign")): continue
        print(mode, k, {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())}, "launches", max(len(v) for v in cs.values()))
PY
