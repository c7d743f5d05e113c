#!/bin/bash
# A/B of two library builds (variants/lib_<name>.so) through the batch tracker's device timeline.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
cp similari_amd/lib/libsimilari_assoc.so /tmp/lib_keep.so
for v in $1; do
  cp variants/lib_$v.so similari_amd/lib/libsimilari_assoc.so
  echo "== $v"
  bash scripts/batch_tracker_timeline.sh ab_$v ${2:-sort 8 500 0 40 0 sync} 2>&1 | grep -v "^W2026" | tail -9
done
cp /tmp/lib_keep.so similari_amd/lib/libsimilari_assoc.so
