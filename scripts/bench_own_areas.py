"""Own-area shares timing: sa_own_areas (one wave per box, boundary integral in f64) against the oracle's or_own_area_shares
(convex-piece decomposition on one host core) on dense random frames.
   python scripts/bench_own_areas.py [n ...]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    sys.path.insert(0, p)
import oracle_lib as O  # noqa: E402
from similari_amd import abi, synth  # noqa: E402
from similari_amd.engine import Engine  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [100, 500, 1000, 4000]
eng = Engine(abi.make_config())
for oriented in (False, True):
    for n in sizes:
        rng = np.random.default_rng(n + oriented)
        side = (n / 1000.0) ** 0.5                      # constant density: the C2 frame (1000 boxes on 1920 x 1080) scaled
        b = synth.dense_boxes(rng, n, (1920.0 * side, 1080.0 * side), oriented=oriented)
        for _ in range(3):
            got = eng.own_areas(b)
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            got = eng.own_areas(b)
        gpu_us = 1e6 * (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        want = O.own_area_shares(b)
        cpu_us = 1e6 * (time.perf_counter() - t0)
        err = float(np.abs(got - want).max())
        assert err < 1e-5, err
        print(json.dumps({"boxes": n, "oriented": oriented, "mean_share": round(float(want.mean()), 4), "max_abs_diff": err,
                          "gpu_us_per_call": round(gpu_us, 1), "oracle_cpu_us": round(cpu_us, 1)}))
eng.close()
