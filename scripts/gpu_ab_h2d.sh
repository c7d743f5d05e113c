#!/bin/bash
# A/B of two library builds (variants/lib_<name>.so) on the ingest-inclusive bench lines.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/ab_h2d; rm -rf $O; mkdir -p $O
cp similari_amd/lib/libsimilari_assoc.so /tmp/lib_keep.so
for r in 1 2; do for v in $1; do
  cp variants/lib_$v.so similari_amd/lib/libsimilari_assoc.so
  for w in ${2:-c2 c3}; do
    timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-oracle > $O/b_${v}_${w}_$r.json 2> $O/b_${v}_${w}_$r.err
    python - $O/b_${v}_${w}_$r.json $v $w <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], sys.argv[3], "resident us", round(1e3 * d["ms_per_step"], 2), "| h2d us", round((d.get("h2d_inclusive") or {}).get("ms_per_step", 0) * 1e3, 2),
      "| device features us", round((d.get("device_features_inclusive") or {}).get("ms_per_step", 0) * 1e3, 2))
PY
  done
done; done
cp /tmp/lib_keep.so similari_amd/lib/libsimilari_assoc.so
