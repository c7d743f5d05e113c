"""The reference's OWN bench scenarios (benches/simple_sort_iou_tracker.rs, simple_sort_iou_tracker_oriented.rs, simple_sort_maha_tracker.rs,
simple_sort_maha_tracker_oriented.rs, simple_visual_sort_tracker.rs, batch_sort_iou_tracker.rs, batch_sort_maha_tracker.rs) through the tracker
facade: object i at (1000 i, 1000 i), 50 x 50 (VisualSORT: 20 x 50), drift 1 px / 0.001, spatio-temporal constraint (1, 1.0), history 10,
max_idle 1; oriented: a random angle in [0, 1) per observation; VisualSORT: Euclidean(10.0), 3 observations, min votes 2, features = 10 index
+ U(-0.01, 0.01) — one whole predict() per iteration, C arrays built outside the timed call like the reference's Vec of observations.
The reference's published figures (assets/benchmarks/benchmarks.md:30-131; another machine: context, not a comparison) ride along.
   python scripts/bench_reference_layouts.py [iters] > profiles/r06_reference_layouts.jsonl"""
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from similari_amd import abi  # noqa: E402
from similari_amd import trackers as TR  # noqa: E402

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
REF_NS = {  # assets/benchmarks/benchmarks.md
    ("sort_iou", 10): 100931, ("sort_iou", 100): 1779434, ("sort_iou", 500): 18705819,
    ("sort_iou_oriented", 10): 108414, ("sort_iou_oriented", 100): 1601062, ("sort_iou_oriented", 500): 18945655,
    ("sort_maha", 10): 105311, ("sort_maha", 100): 1696943, ("sort_maha", 500): 18233557,
    ("sort_maha_oriented", 10): 111778, ("sort_maha_oriented", 100): 1567771, ("sort_maha_oriented", 500): 17762559,
    ("batch_sort_iou", 10): 106876, ("batch_sort_iou", 100): 1616542, ("batch_sort_iou", 500): 20454230,
    ("batch_sort_maha", 10): 114592, ("batch_sort_maha", 100): 1533445, ("batch_sort_maha", 500): 18270742,
    ("visual_sort", 10, 128): 356237, ("visual_sort", 10, 512): 447903, ("visual_sort", 10, 2048): 767031,
    ("visual_sort", 50, 128): 1923861, ("visual_sort", 50, 512): 2249694, ("visual_sort", 50, 2048): 4563691,
    ("visual_sort", 100, 128): 3807716, ("visual_sort", 100, 256): 4717401, ("visual_sort", 100, 512): 5775469,
    ("visual_sort", 100, 1024): 7497783, ("visual_sort", 100, 2048): 10527237,
}


class BoxGen2:   # examples.rs:188-248
    def __init__(self, rng, n, w, h, pos_drift=1.0, box_drift=0.001):
        i = np.arange(n, dtype=np.float32)
        self.x, self.y = 1000.0 * i, 1000.0 * i
        self.w, self.h = np.full(n, w, np.float32), np.full(n, h, np.float32)
        self.rng, self.pd, self.bd = rng, pos_drift, box_drift

    def next(self):
        n = len(self.x)
        self.x = self.x + self.rng.uniform(-self.pd, self.pd, n).astype(np.float32)
        self.y = self.y + self.rng.uniform(-self.pd, self.pd, n).astype(np.float32)
        self.w = np.maximum(self.w + self.rng.uniform(-self.bd, self.bd, n).astype(np.float32), 1.0)
        self.h = np.maximum(self.h + self.rng.uniform(-self.bd, self.bd, n).astype(np.float32), 1.0)
        return self.x, self.y, self.w, self.h   # left, top, width, height


def run(name, objects, feature_len=0, device_upkeep=True):
    rng = np.random.default_rng(7)
    visual = name == "visual_sort"
    oriented = name.endswith("oriented")
    batch = name.startswith("batch")
    maha = "maha" in name
    cons = TR.SpatioTemporalConstraints().add_constraints([(1, 1.0)])
    method = TR.PositionalMetricType.maha() if maha else TR.PositionalMetricType.iou(0.3)
    if visual:
        opts = (TR.VisualSortOptions().positional_metric(TR.PositionalMetricType.iou(0.3)).visual_metric(TR.VisualSortMetricType.euclidean(10.0))
                .visual_max_observations(3).spatio_temporal_constraints(cons).visual_minimal_own_area_percentage_use(0.5)
                .visual_minimal_own_area_percentage_collect(0.6).visual_min_votes(2))
        trk = TR.VisualSort(opts=opts, feature_len=feature_len, device_upkeep=device_upkeep)
    elif batch:
        trk = TR.BatchSort(bbox_history=10, max_idle_epochs=1, method=method, min_confidence=0.05, spatio_temporal_constraints=cons, device_upkeep=device_upkeep)
    else:
        trk = TR.Sort(bbox_history=10, max_idle_epochs=1, method=method, min_confidence=0.05, spatio_temporal_constraints=cons, device_upkeep=device_upkeep)
    gen = BoxGen2(rng, objects, 20.0 if visual else 50.0, 50.0)
    lib = trk.lib
    out = (abi.sa_sort_track * objects)()
    times = []
    for it in range(ITERS + 20):
        left, top, w, h = gen.next()
        arr = (abi.sa_observation * objects)()
        keep = []
        feats = (rng.uniform(-0.01, 0.01, (objects, feature_len)).astype(np.float32) + 10.0 * np.arange(objects, dtype=np.float32)[:, None]) if visual else None
        for i in range(objects):
            o = arr[i]
            b = o.bbox
            b.xc, b.yc, b.aspect, b.height, b.confidence = float(left[i] + w[i] / 2), float(top[i] + h[i] / 2), float(w[i] / h[i]), float(h[i]), 1.0
            if oriented:
                b.has_angle, b.angle = 1, float(rng.uniform(0.0, 1.0))
            o.feature_quality = 1.0 if visual else float("nan")
            o.own_area = float("nan")
            o.has_custom_object_id, o.custom_object_id = (1, 0) if visual else (0, 0)
            if visual:
                o.feature = feats[i].ctypes.data_as(C.POINTER(C.c_float))
        if batch:
            ids = (C.c_uint64 * 1)(0)
            counts = (C.c_uint32 * 1)(objects)
            pa = (C.POINTER(abi.sa_observation) * 1)(C.cast(arr, C.POINTER(abi.sa_observation)))
            po = (C.POINTER(abi.sa_sort_track) * 1)(C.cast(out, C.POINTER(abi.sa_sort_track)))
            t0 = time.perf_counter()
            rc = lib.sa_tracker_predict_batch(trk.h, 1, ids, counts, pa, po)
        else:
            t0 = time.perf_counter()
            rc = lib.sa_tracker_predict(trk.h, 0, objects, arr, out)
        dt = time.perf_counter() - t0
        assert rc == 0, lib.sa_tracker_last_error(trk.h)
        if it >= 20:
            times.append(dt)
    kept = sum(1 for i in range(objects) if out[i].length > ITERS)
    active = trk.active_tracks()
    trk.close()
    key = (name, objects, feature_len) if visual else (name, objects)
    ns = 1e9 * float(np.median(times))
    return {"bench": name, "objects": objects, "feature_len": feature_len or None, "upkeep": "device" if device_upkeep else "host",
            "ns_per_predict_median": round(ns), "ns_per_predict_p10": round(1e9 * float(np.percentile(times, 10))), "ns_per_predict_p90": round(1e9 * float(np.percentile(times, 90))),
            "fps": round(1e9 / ns), "tracks_kept_through_the_run": kept, "active_tracks": active,
            "reference_ns_per_iter_benchmarks_md": REF_NS.get(key), "speedup_vs_published_other_hardware": round(REF_NS[key] / ns, 1) if key in REF_NS else None}


if __name__ == "__main__":
    for name in ("sort_iou", "sort_iou_oriented", "sort_maha", "sort_maha_oriented", "batch_sort_iou", "batch_sort_maha"):
        for objects in (10, 100, 500):
            print(json.dumps(run(name, objects)), flush=True)
    for objects in (10, 50, 100):
        for d in (128, 512, 2048):
            print(json.dumps(run("visual_sort", objects, d)), flush=True)
    print(json.dumps(run("visual_sort", 100, 256)), flush=True)
    print(json.dumps(run("visual_sort", 100, 1024)), flush=True)
    print(json.dumps(run("sort_iou", 500, device_upkeep=False)), flush=True)
    print(json.dumps(run("visual_sort", 100, 512, device_upkeep=False)), flush=True)
