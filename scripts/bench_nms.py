"""NMS timing: sa_nms (GPU pair tests + greedy sweep) against the oracle's or_nms on the host, on the reference bench's
distribution scaled up (benches/nms.rs: BoxGen2 objects; here dense random boxes with near-duplicates).
   python scripts/bench_nms.py [n ...]"""
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
        b = synth.dense_boxes(rng, n, (1920.0, 1080.0), oriented=oriented)
        dup = synth.jitter_boxes(rng, b[: n // 2], 3.0, size_rel=0.05, angle_sigma=0.05 if oriented else 0.0)
        b = np.concatenate([b, dup])[rng.permutation(n + n // 2)][:n]
        for _ in range(3):
            got = eng.nms(b, None, 0.5, None)
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            got = eng.nms(b, None, 0.5, None)
        gpu_us = 1e6 * (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        want = O.nms(b, None, 0.5, None)
        cpu_us = 1e6 * (time.perf_counter() - t0)
        assert np.array_equal(got, want)
        print(json.dumps({"boxes": n, "oriented": oriented, "kept": int(len(got)), "gpu_us_per_call": round(gpu_us, 1),
                          "oracle_cpu_us": round(cpu_us, 1), "pairs_per_s_gpu": round(n * (n - 1) / 2 / (gpu_us * 1e-6))}))
eng.close()
