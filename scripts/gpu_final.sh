#!/bin/bash
# Round-end visit: the whole GPU suite, smoke(), then the profile collection.   scripts/gpu_final.sh <tag>
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r02_z}
bash scripts/profile_round2.sh $TAG
O=$PWD/gpurun_out/$TAG
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -n 4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log; tail -n 2 $O/smoke.log
