"""Probe of the k-split main loop (gemm plans 9, 10, 13) through sa_feature_distance_matrix: answers against an f64 reference, then
microseconds per launch next to the staged plans on the C2 family of shapes."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from similari_amd import abi
from similari_amd.engine import Engine

plans = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,7,9,10,13".split(","))]
rng = np.random.default_rng(0)
for (n, t, d) in [(300, 333, 512), (129, 70, 96), (64, 64, 32), (1000, 1000, 512)]:
    a = rng.standard_normal((n, d)).astype(np.float32)
    b = rng.standard_normal((t, d)).astype(np.float32)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    ref = (a64 @ b64.T) / np.sqrt((a64 * a64).sum(1)[:, None] * (b64 * b64).sum(1)[None, :])
    for plan in plans:
        eng = Engine(abi.make_config(gemm_plan=plan))
        out, _ = eng.distance_matrix("cosine", a, b)
        eng.close()
        print(json.dumps({"check": [n, t, d], "plan": plan, "max_err": float(np.abs(out - ref).max())}), flush=True)
shapes = {"c2": (1000, 1000, 512), "c2t": (1000, 1500, 512), "c2k3": (1000, 3000, 512), "c2x2": (1000, 2000, 512), "d128": (1000, 1000, 128), "d2048": (1000, 1000, 2048)}
for name, (n, t, d) in shapes.items():
    a = rng.standard_normal((n, d)).astype(np.float32)
    b = rng.standard_normal((t, d)).astype(np.float32)
    for plan in plans:
        eng = Engine(abi.make_config(gemm_plan=plan))
        best = 1e9
        for rep in range(3):
            _, ms = eng.distance_matrix("cosine", a, b, iters=100, want_out=False)
            best = min(best, 1e3 * ms / 100)
        eng.close()
        tf = 2.0 * n * t * d / (best * 1e-6) / 1e12
        print(json.dumps({"shape": name, "plan": plan, "us": round(best, 2), "frac": round(tf / 157.3, 3)}), flush=True)
