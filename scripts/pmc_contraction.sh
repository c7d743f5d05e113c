cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/pmch2
for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $pass | md5sum | cut -c1-6)
  (cd /tmp && timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/pmch2/$tag -o p -- python $OLDPWD/bench.py --workload c5 --flags ${FLAGS:-64} --steps 6 --warmup 2 --no-cpu-baseline --profile-iters 2 > $OLDPWD/gpurun_out/pmch2/$tag.log 2>&1)
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmch2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "visual_cos" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()): print(k, round(sum(v)/len(v)))
PY
