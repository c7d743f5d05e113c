#!/bin/bash
# (measurement only) the fused first phase with phases of its positional tiles cut out: what each phase costs the launch
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; : > gpurun_out/pos_skip.txt
for sk in 0 2 3 4; do
  SA_EXTRA_FLAGS="-DSA_POS_SKIP=$sk" python -m similari_amd.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  for rep in 1 2; do
  timeout 300 python bench.py --workload ${1:-c2} --no-cpu-baseline --no-oracle --no-h2d 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip=$sk', d['ms_per_step'], {k:round(x['avg_us'],2) for k,x in d['kernels'].items()})" | tee -a gpurun_out/pos_skip.txt
  done
  O=$PWD/gpurun_out/pos_skip_pmc; rm -rf $O
  (cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O -o p -- python $OLDPWD/bench.py --workload ${1:-c2} --steps 20 --warmup 5 --no-cpu-baseline --no-oracle --no-h2d > /dev/null 2>&1)
  python - $O $sk <<'PY' | tee -a gpurun_out/pos_skip.txt
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_frame_visual" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("skip=" + sys.argv[2], {c: round(sum(v) / len(v)) for c, v in sorted(acc.items())})
PY
done
