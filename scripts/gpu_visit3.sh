#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_c; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -x -q -p no:cacheprovider -k "euclid or pipelined or associate_batch" > $O/pytest_first.log 2>&1; echo "first exit $?" >> $O/pytest_first.log; tail -15 $O/pytest_first.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -n 12 $O/pytest.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h = d.get("h2d_inclusive") or {}
    print(sys.argv[2], {k: d.get(k) for k in ("ms_per_step", "value_h2d", "match_vs_oracle")}, {k: round(v["avg_us"], 2) for k, v in d.get("kernels", {}).items()},
          {k: h.get(k) for k in ("ms_per_step", "synchronous_ms_per_step", "host_us_in_submit", "host_us_in_wait")}, (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for v in default sdma; do
  if [ $v = sdma ]; then export SA_INGEST=sdma; else unset SA_INGEST; fi
  timeout 300 python bench.py --no-cpu-baseline --no-oracle > $O/bench_c2_$v.json 2> $O/bench_c2_$v.err; show $O/bench_c2_$v.json "c2 ingest=$v"
done
unset SA_INGEST
for b in 24 96; do SA_INGEST_BLOCKS=$b timeout 300 python bench.py --no-cpu-baseline --no-oracle > $O/bench_c2_b$b.json 2> $O/bench_c2_b$b.err; show $O/bench_c2_b$b.json "c2 ingest blocks=$b"; done
timeout 300 python bench.py --workload c2e --no-cpu-baseline > $O/bench_c2e.json 2> $O/bench_c2e.err; show $O/bench_c2e.json "c2e mfma"
SA_EUCLID=valu timeout 300 python bench.py --workload c2e --no-cpu-baseline > $O/bench_c2e_valu.json 2> $O/bench_c2e_valu.err; show $O/bench_c2e_valu.json "c2e valu"
timeout 300 python bench.py --workload c2e --flags 32 --no-cpu-baseline --no-oracle > $O/bench_c2e_sep.json 2> $O/bench_c2e_sep.err; show $O/bench_c2e_sep.json "c2e mfma separate"
timeout 300 python bench.py --workload c3 --no-cpu-baseline --no-oracle > $O/bench_c3.json 2> $O/bench_c3.err; show $O/bench_c3.json "c3"
timeout 600 python bench.py --workload c5 --no-cpu-baseline --no-oracle > $O/bench_c5.json 2> $O/bench_c5.err; show $O/bench_c5.json "c5"
echo DONE
