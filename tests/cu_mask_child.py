"""Child process of tests/test_gpu_progress.py: the many-workgroup assignment tail on a device that may have been masked down to a
couple of compute units by the parent (ROC_GLOBAL_CU_MASK / HSA_CU_MASK are read when the runtime starts: a process of its own).
Prints one JSON line: the time of a fixed compute-bound launch (how much of the device this process really has) and, per frame, whether
the engine's answer is the oracle's."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import oracle_lib as O  # noqa: E402
from similari_amd import abi, synth  # noqa: E402
from similari_amd.engine import Engine  # noqa: E402


def total_gain(ids, q, track_ids, thr_q):
    col = {int(t): j for j, t in enumerate(track_ids)}
    return sum(int(q[i, col[int(t)]]) - thr_q for i, t in enumerate(ids) if t)


def main():
    out = {"frames": []}
    cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
    eng = Engine(cfg)
    try:
        # how much of the device is there: 2000 x 2000 x 512 cosine distances, the contraction alone
        rng = np.random.default_rng(5)
        a = rng.standard_normal((2000, 512)).astype(np.float32)
        b = rng.standard_normal((2000, 512)).astype(np.float32)
        eng.distance_matrix("cosine", a, b, iters=3)
        out["gemm_ms"] = eng.distance_matrix("cosine", a, b, iters=10)[1] / 10.0
        # (a) a crowd beyond the one-workgroup tail: hundreds of components, dozens of mid-sized ones, a few for the dense solver
        # (b) many rows: 16 row workgroups of 256 — more than a masked device holds at once
        for name, n, t, canvas, sigma, thr in (("crowd 1000 x 2500", 1000, 2500, (1920.0, 1080.0), 2.0, 0.3), ("crowd 1500 x 1300", 1500, 1300, (1920.0, 1080.0), 8.0, 0.3),
                                               ("rows 4096 x 4096", 4096, 4096, (9000.0, 6000.0), 3.0, 0.3)):
            sc = synth.sort_scene(np.random.default_rng(n + 7 * t), t, n, canvas=canvas, pos_sigma=sigma)
            c2 = abi.make_config(positional="iou", positional_threshold=thr, max_idle_epochs=5)
            tracks = abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"])
            det = abi.make_detections(sc["det_boxes"])
            ref = O.associate(c2, tracks, 1, det)
            e2 = Engine(c2)
            try:
                e2.upsert(0, tracks)
                t0 = time.perf_counter()
                ids, votes = e2.associate(0, 1, det)
                ids2, votes2 = e2.associate(0, 1, det)     # the tail leaves its state clean: the same frame again
                dt = time.perf_counter() - t0
            finally:
                e2.close()
            thr_q = int(O.lib().or_quantise(c2.positional_threshold))
            same_gain = total_gain(ids, ref["quantised"], sc["track_ids"], thr_q) == total_gain(ref["track_id"], ref["quantised"], sc["track_ids"], thr_q)
            out["frames"].append({"frame": name, "ids_match": bool(np.array_equal(ids, ref["track_id"])), "votes_match": bool(np.array_equal(votes, ref["voting_type"])),
                                  "same_total_gain": bool(same_gain), "repeat_matches": bool(np.array_equal(ids, ids2) and np.array_equal(votes, votes2)),
                                  "matched": int((ids != 0).sum()), "two_frames_ms": 1e3 * dt})
    finally:
        eng.close()
    print("CU-MASK-CHILD " + json.dumps(out))


if __name__ == "__main__":
    main()
