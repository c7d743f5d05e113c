#!/bin/bash
# PMC pass for the contraction kernel: MFMA busy cycles vs GPU active cycles (separate run, --pmc only).
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc
rocprofv3 -L 2>/dev/null | grep -iE "MFMA|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_WAIT_INST|LDS_BANK|FETCH_SIZE|WRITE_SIZE|SQ_ACTIVE_INST" | head -40 > gpurun_out/pmc/counters.txt
SHAPE=${1:-sq4k}; PLAN=${2:-0}
(cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/pmc/a -o p -- python $OLDPWD/scripts/gemm_bench.py $SHAPE $PLAN > $OLDPWD/gpurun_out/pmc/a.log 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/pmc/b -o p -- python $OLDPWD/scripts/gemm_bench.py $SHAPE $PLAN > $OLDPWD/gpurun_out/pmc/b.log 2>&1)
python - <<'PY'
import csv, glob, collections
for tag in "ab":
    for f in glob.glob(f"gpurun_out/pmc/{tag}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        for k, d in agg.items():
            if "cosine" in k or "euclid" in k: print(tag, k, {c: v for c, v in d.items()})
PY
cat gpurun_out/pmc/counters.txt | head -30
