#!/bin/bash
# A/B of two builds of the library (variants/lib_<name>.so, built here beforehand) on ONE box: bench lines of the listed workloads, alternating.
#   bash scripts/gpu_ab.sh <tag> "<variant names>" "<workloads>" [rounds]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-ab}; VARS=${2:-"a b"}; WL=${3:-"c4 c3"}; R=${4:-2}
O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cp similari_amd/lib/libsimilari_assoc.so $O/lib_keep.so
for r in $(seq 1 $R); do for v in $VARS; do
  cp variants/lib_$v.so similari_amd/lib/libsimilari_assoc.so
  for w in $WL; do
    timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-oracle > $O/b_${v}_${w}_$r.json 2> $O/b_${v}_${w}_$r.err
    python - $O/b_${v}_${w}_$r.json $v $w <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], sys.argv[3], "us/frame", round(1e3 * d["ms_per_step"], 2), {k: round(v.get("avg_us", 0), 2) if isinstance(v, dict) else v for k, v in (d.get("kernels") or {}).items()})
except Exception as e:
    print("unreadable", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
  done
done; done
cp $O/lib_keep.so similari_amd/lib/libsimilari_assoc.so; rm -f $O/lib_keep.so
