#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_r0; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --durations=8 > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -n 16 $O/pytest.log
echo DONE
