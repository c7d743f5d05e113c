"""Microbenchmark of the feature contraction alone (sa_feature_distance_matrix): TFLOP/s per tile plan."""
import os, sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from similari_amd import abi
from similari_amd.engine import Engine

shapes = {"c2": (1000, 1000, 512), "c2k3": (1000, 3000, 512), "c5": (2000, 5000, 4096), "sq4k": (4096, 4096, 4096)}
which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c2", "c5"]
plans = sys.argv[2].split(",") if len(sys.argv) > 2 else ["auto"]
kind = sys.argv[3] if len(sys.argv) > 3 else "cosine"
rng = np.random.default_rng(0)
for name in which:
    n, t, d = shapes[name]
    a = rng.standard_normal((n, d)).astype(np.float32)
    b = rng.standard_normal((t, d)).astype(np.float32)
    for plan in plans:
        eng = Engine(abi.make_config(gemm_plan=None if plan == "auto" else int(plan)))
        iters = 50 if name.startswith("c2") else 10
        _, ms = eng.distance_matrix(kind, a, b, iters=iters, want_out=False)
        eng.close()
        us = 1e3 * ms / iters
        tf = 2.0 * n * t * d / (us * 1e-6) / 1e12
        print(json.dumps({"shape": name, "plan": plan, "kind": kind, "us": round(us, 2), "tflops": round(tf, 2), "frac_of_157.3": round(tf / 157.3, 3)}))
