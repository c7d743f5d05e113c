#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_pp; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 300 python scripts/debug_big_visual.py 2>&1 | tail -40
echo "--- SA_TAIL=general on 600"; SA_TAIL=general timeout 300 python scripts/debug_big_visual.py 2>&1 | tail -12
echo DONE
