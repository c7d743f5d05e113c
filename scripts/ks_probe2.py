"""Probe of the direct-load loop of the wider tiles (gemm plans 15-17) against the staged plans 0 / 5 / 6."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from similari_amd import abi
from similari_amd.engine import Engine
plans = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,5,6,15,16,17".split(","))]
rng = np.random.default_rng(0)
for (n, t, d) in [(300, 333, 512), (129, 70, 96), (64, 64, 32)]:
    a = rng.standard_normal((n, d)).astype(np.float32); b = rng.standard_normal((t, d)).astype(np.float32)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    ref = (a64 @ b64.T) / np.sqrt((a64 * a64).sum(1)[:, None] * (b64 * b64).sum(1)[None, :])
    for plan in plans:
        eng = Engine(abi.make_config(gemm_plan=plan)); out, _ = eng.distance_matrix("cosine", a, b); eng.close()
        print(json.dumps({"check": [n, t, d], "plan": plan, "max_err": float(np.abs(out - ref).max())}), flush=True)
shapes = {"c2bish": (4096, 2048, 512), "c5": (2000, 5000, 4096), "c2k3": (1000, 3000, 512), "sq4k": (4096, 4096, 1024)}
for name, (n, t, d) in shapes.items():
    a = rng.standard_normal((n, d)).astype(np.float32); b = rng.standard_normal((t, d)).astype(np.float32)
    for plan in plans:
        eng = Engine(abi.make_config(gemm_plan=plan)); best = 1e9
        it = 10 if name == "c5" else 50
        for rep in range(3):
            _, ms = eng.distance_matrix("cosine", a, b, iters=it, want_out=False); best = min(best, 1e3 * ms / it)
        eng.close()
        print(json.dumps({"shape": name, "plan": plan, "us": round(best, 2), "frac": round(2.0 * n * t * d / (best * 1e-6) / 1e12 / 157.3, 3)}), flush=True)
