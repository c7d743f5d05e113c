cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
timeout 900 bash scripts/gpu_visit.sh r03_ab all c4 c5 sdt 2>&1 | tail -14
timeout 200 python scripts/bench_tracker.py 1000 512 30 2>/dev/null | head -2
