#!/bin/bash
# contraction experiments: stand-alone plans at the C2 shape (+ correctness of the new plan against numpy)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r05_gemm}; O=$PWD/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; }
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
from similari_amd import abi
from similari_amd.engine import Engine
rng = np.random.default_rng(1)
for (n, t, d) in ((1000, 1000, 512), (333, 777, 96), (64, 64, 32), (65, 129, 64)):
    a = rng.standard_normal((n, d)).astype(np.float32); b = rng.standard_normal((t, d)).astype(np.float32)
    ref = (a.astype(np.float64) @ b.astype(np.float64).T) / np.sqrt((a.astype(np.float64) ** 2).sum(1)[:, None] * (b.astype(np.float64) ** 2).sum(1)[None, :])
    for plan in (1, 9):
        eng = Engine(abi.make_config(gemm_plan=plan))
        out, ms = eng.distance_matrix("cosine", a, b, iters=1)
        eng.close()
        print("plan", plan, (n, t, d), "max abs err", float(np.abs(out - ref).max()))
PY
python scripts/gemm_bench.py c2,c2k3 1,2,7,9
echo DONE
