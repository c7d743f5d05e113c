"""End-to-end tracker loop (facade predict() per frame): host upkeep (Kalman + bank on the host, sa_tracks_upsert every frame)
against device upkeep (sa_tracks_apply).  Times the C call only (observations are built once).
   python scripts/bench_tracker.py [n_objects] [feature_len] [frames]"""
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from similari_amd import abi, synth  # noqa: E402
from similari_amd import trackers as TR  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rng = np.random.default_rng(0)
ident = synth.reid_identities(rng, n, d)
world0 = synth.dense_boxes(rng, n, (1920.0, 1080.0))


def u2d(b):
    return TR.Universal2DBox(float(b["xc"]), float(b["yc"]), None, float(b["aspect"]), float(b["height"]), float(b["confidence"]))


def run(kind, device_upkeep):
    if kind == "visual":
        opts = (TR.VisualSortOptions().max_idle_epochs(3).kept_history_length(3).visual_metric(TR.VisualSortMetricType.cosine(0.2))
                .positional_metric(TR.PositionalMetricType.iou(0.3)).visual_minimal_track_length(1).visual_max_observations(3))
        trk = TR.VisualSort(opts=opts, feature_len=d, device_upkeep=device_upkeep)
    else:
        trk = TR.Sort(bbox_history=3, max_idle_epochs=3, device_upkeep=device_upkeep)
    world = world0.copy()
    r = np.random.default_rng(1)
    times = []
    for f in range(frames):
        world = synth.jitter_boxes(r, world, 2.0)
        feats = synth.observe(r, ident, 0.01)
        if kind == "visual":
            items = [TR.VisualSortObservation(feats[k], 0.9, u2d(world[k]), None) for k in range(n)]
        else:
            items = [(u2d(world[k]), None) for k in range(n)]
        keep = []
        arr = trk._obs_array(items, keep)
        out = (abi.sa_sort_track * n)()
        t0 = time.perf_counter()
        rc = trk.lib.sa_tracker_predict(trk.h, 0, n, arr, out)
        times.append(time.perf_counter() - t0)
        assert rc == 0
    matched = sum(1 for i in range(n) if out[i].length > 1)
    trk.close()
    return 1e3 * float(np.median(times[3:])), matched


for kind in ("sort", "visual"):
    for dev in (False, True):
        ms, matched = run(kind, dev)
        print(json.dumps({"tracker": kind, "objects": n, "feature_len": d if kind == "visual" else 0, "upkeep": "device" if dev else "host",
                          "ms_per_frame_median": round(ms, 3), "tracks_continued_last_frame": matched}))
