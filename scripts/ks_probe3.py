import sys, json
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from similari_amd import abi
from similari_amd.engine import Engine
rng = np.random.default_rng(0)
for name, (n, t, d) in {"c5T": (5000, 2000, 4096), "c5": (2000, 5000, 4096)}.items():
    a = rng.standard_normal((n, d)).astype(np.float32); b = rng.standard_normal((t, d)).astype(np.float32)
    for plan in [0, 5, 6, 15, 16, 17]:
        eng = Engine(abi.make_config(gemm_plan=plan)); best = 1e9
        for rep in range(3):
            _, ms = eng.distance_matrix("cosine", a, b, iters=10, want_out=False); best = min(best, 1e3 * ms / 10)
        eng.close()
        print(json.dumps({"shape": name, "plan": plan, "us": round(best, 2), "frac": round(2.0 * n * t * d / (best * 1e-6) / 1e12 / 157.3, 3)}), flush=True)
