#!/bin/bash
# Completion words against completion signals through the batch facade, alternating on one box.  The two builds are made here beforehand:
#   python -c "from similari_amd import build; build.build_lib(force=True)"; cp similari_amd/lib/libsimilari_assoc.so variants/lib_words.so
#   SA_EXTRA_FLAGS=-DSA_FORCE_SIGNAL_COMPLETION python -c "from similari_amd import build; build.build_lib(force=True)"; cp ... variants/lib_signal.so
# (scripts/gpu_ab_timeline.sh "words signal" reads the same two builds' device timelines.)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_trackers.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3
cp similari_amd/lib/libsimilari_assoc.so /tmp/lib_keep.so
for r in 1 2; do for v in words signal; do
  cp variants/lib_$v.so similari_amd/lib/libsimilari_assoc.so
  for cfg in "sort 8 500 0 40 0" "sort 64 500 0 30 0" "visual 8 1000 512 24 0"; do
    timeout 200 python scripts/bench_batch_tracker.py $cfg sync device 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   $v | $cfg |', d['us_per_predict_median'], d['us_per_predict_min'])"
  done
done; done
cp /tmp/lib_keep.so similari_amd/lib/libsimilari_assoc.so
