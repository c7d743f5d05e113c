#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r05_k}; O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; }
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 -p no:cacheprovider --maxfail=10 -rf -s -k "$2" > $O/pytest.log 2>&1; echo "exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|bookkeeping|Error" $O/pytest.log | cut -c1-400 | tail -20
echo DONE
