"""BatchSort / BatchVisualSort::predict through the facade (device upkeep): S scenes of n objects per call, the C arrays built before the
timed loop and the C calls back to back (see scripts/bench_tracker.py).  BASELINE C3's shape by default: 8 scenes x 500 objects.
   python scripts/bench_batch_tracker.py [sort|visual] [scenes] [objects] [feature_len] [frames]"""
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from similari_amd import abi, synth  # noqa: E402
from similari_amd import trackers as TR  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "sort"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 500
d = int(sys.argv[4]) if len(sys.argv) > 4 else 256
frames = int(sys.argv[5]) if len(sys.argv) > 5 else 30
rng = np.random.default_rng(0)
if kind == "visual":
    opts = (TR.VisualSortOptions().max_idle_epochs(3).kept_history_length(3).visual_metric(TR.VisualSortMetricType.cosine(0.2))
            .positional_metric(TR.PositionalMetricType.iou(0.3)).visual_minimal_track_length(1).visual_max_observations(3))
    trk = TR.BatchVisualSort(opts=opts, feature_len=d, device_upkeep=True)
else:
    trk = TR.BatchSort(bbox_history=3, max_idle_epochs=3, device_upkeep=True)
lib = trk.lib
worlds = [synth.dense_boxes(rng, n, (1920.0, 1080.0)) for _ in range(S)]
idents = [synth.reid_identities(rng, n, d) for _ in range(S)] if kind == "visual" else None
keep, calls = [], []
for f in range(frames):
    arrs = []
    for s in range(S):
        worlds[s] = synth.jitter_boxes(rng, worlds[s], 2.0)
        boxes = [TR.Universal2DBox(float(b["xc"]), float(b["yc"]), None, float(b["aspect"]), float(b["height"]), float(b["confidence"])) for b in worlds[s]]
        if kind == "visual":
            feats = synth.observe(rng, idents[s], 0.01)
            keep.append(feats)
            items = [TR.VisualSortObservation(feats[k], 0.9, boxes[k], None) for k in range(n)]
        else:
            items = [(boxes[k], None) for k in range(n)]
        arrs.append(trk._obs_array(items, keep))
    outs = [(abi.sa_sort_track * n)() for _ in range(S)]
    ids = (C.c_uint64 * S)(*range(S))
    counts = (C.c_uint32 * S)(*([n] * S))
    pa = (C.POINTER(abi.sa_observation) * S)(*[C.cast(a, C.POINTER(abi.sa_observation)) for a in arrs])
    po = (C.POINTER(abi.sa_sort_track) * S)(*[C.cast(o, C.POINTER(abi.sa_sort_track)) for o in outs])
    calls.append((arrs, outs, ids, counts, pa, po))
times = []
for (arrs, outs, ids, counts, pa, po) in calls:
    t0 = time.perf_counter()
    rc = lib.sa_tracker_predict_batch(trk.h, S, ids, counts, pa, po)
    times.append(time.perf_counter() - t0)
    assert rc == 0, lib.sa_tracker_last_error(trk.h)
cont = sum(1 for o in calls[-1][1] for i in range(n) if o[i].length > 1)
trk.close()
print(json.dumps({"tracker": "Batch" + ("VisualSort" if kind == "visual" else "Sort"), "scenes": S, "objects_per_scene": n, "feature_len": d if kind == "visual" else 0,
                  "upkeep": "device", "ms_per_predict_median": round(1e3 * float(np.median(times[3:])), 3),
                  "us_per_scene": round(1e6 * float(np.median(times[3:])) / S, 1), "tracks_continued_last_frame": cont}))
