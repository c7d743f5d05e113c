"""End-to-end tracker loop (facade predict() per frame): host upkeep (Kalman + bank on the host, sa_tracks_upsert every frame)
against device upkeep (sa_tracks_apply).  The observations of every frame are built BEFORE the timed loop and the C calls run back
to back (the way the reference's criterion benches call predict(), benches/simple_visual_sort_tracker.rs) — a loop that leaves the
GPU idle for milliseconds between frames measures the wake-up of an idle queue (~100 us per frame on this stack), not the tracker.
VisualSORT features three ways: one host array per observation (`rows`: the facade gathers them), one pinned N x D block per frame
from sa_host_alloc (`pinned`: read in place over the link), one N x D block per frame in device memory registered with
sa_device_block_register (`device`: the ReID model's output buffer on the same GPU, read where it lies).
`churn` > 0: that fraction of the objects leaves and as many enter EVERY frame (max_idle_epochs 3): the departed tracks linger in the
table until they are wasted, so the table holds more rows than the frame has detections — T > 1024 at 1000 objects, a tracker loop's
normal state (the many-workgroup assignment tail, vote words beyond the one-workgroup tail's 1024 x 1024).
   python scripts/bench_tracker.py [n_objects] [feature_len] [frames]"""
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

torch.zeros(1, device="cuda:0")  # torch's HIP context first (it stands in for the detector / ReID model that owns the feature buffers)

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from similari_amd import abi, synth  # noqa: E402
from similari_amd import trackers as TR  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rng = np.random.default_rng(0)
pool = n + frames * max(1, n // 20) + 8          # identities: the n of frame 0 + everything a churned loop brings in
ident = synth.reid_identities(rng, pool, d)
world0 = synth.dense_boxes(rng, pool, (1920.0, 1080.0))


def u2d(b):
    return TR.Universal2DBox(float(b["xc"]), float(b["yc"]), None, float(b["aspect"]), float(b["height"]), float(b["confidence"]))


def run(kind, device_upkeep, feats_mode="rows", churn=0.0):
    if kind == "visual":
        opts = (TR.VisualSortOptions().max_idle_epochs(3).kept_history_length(3).visual_metric(TR.VisualSortMetricType.cosine(0.2))
                .positional_metric(TR.PositionalMetricType.iou(0.3)).visual_minimal_track_length(1).visual_max_observations(3))
        trk = TR.VisualSort(opts=opts, feature_len=d, device_upkeep=device_upkeep)
    else:
        trk = TR.Sort(bbox_history=3, max_idle_epochs=3, device_upkeep=device_upkeep)
    lib = trk.lib
    world = world0.copy()
    r = np.random.default_rng(1)
    active = np.arange(n)      # which identity each of the n detections of a frame belongs to
    fresh = n
    keep, arrs, blocks = [], [], []
    dev = None
    if kind == "visual" and feats_mode == "device":
        dev = torch.empty((frames, n, d), dtype=torch.float32, device="cuda:0")
        lib.sa_device_block_register(C.c_void_p(dev.data_ptr()), dev.numel() * 4, 0)
    for f in range(frames):
        world = synth.jitter_boxes(r, world, 2.0)
        if churn > 0.0 and f > 0:
            k_out = max(1, int(churn * n))
            gone = r.choice(n, k_out, replace=False)
            active[gone] = np.arange(fresh, fresh + k_out)
            fresh += k_out
        feats = synth.observe(r, ident[active], 0.01)
        if kind == "visual":
            if feats_mode == "pinned":
                p = lib.sa_host_alloc(n * d * 4)
                blk = np.frombuffer((C.c_char * (n * d * 4)).from_address(p), dtype=np.float32).reshape(n, d)
                blk[...] = feats
                blocks.append(p)
                feats = blk
            items = [TR.VisualSortObservation(feats[k], 0.9, u2d(world[active[k]]), None) for k in range(n)]
        else:
            items = [(u2d(world[active[k]]), None) for k in range(n)]
        arr = trk._obs_array(items, keep)
        if dev is not None:
            dev[f].copy_(torch.from_numpy(feats))
            base = dev.data_ptr() + f * n * d * 4
            for k in range(n):
                arr[k].feature = C.cast(C.c_void_p(base + k * d * 4), C.POINTER(C.c_float))
        arrs.append(arr)
    if dev is not None:
        torch.cuda.synchronize()
    out = (abi.sa_sort_track * n)()
    times = []
    for f in range(frames):  # back to back: nothing but the C call inside the loop
        t0 = time.perf_counter()
        rc = lib.sa_tracker_predict(trk.h, 0, n, arrs[f], out)
        times.append(time.perf_counter() - t0)
        assert rc == 0, lib.sa_tracker_last_error(trk.h)
    matched = sum(1 for i in range(n) if out[i].length > 1)
    table_rows = C.c_uint32()
    lib.sa_tracks_count(lib.sa_tracker_engine(trk.h), 0, C.byref(table_rows))
    trk.close()
    if dev is not None:
        lib.sa_device_block_unregister(C.c_void_p(dev.data_ptr()))
    for p in blocks:
        lib.sa_host_free(p)
    return 1e3 * float(np.median(times[3:])), matched, int(table_rows.value)


import os  # noqa: E402

only = os.environ.get("SA_BENCH_TRACKER_ONLY")   # e.g. "visual,device,0.0" (tracker, features, churn): that one device-upkeep run only
if only:
    k_, m_, c_ = only.split(",")
    ms, matched, rows = run(k_, True, m_, float(c_))
    print(json.dumps({"tracker": k_, "features": m_, "churn_per_frame": float(c_), "ms_per_frame_median": round(ms, 3), "table_rows_last_frame": rows}), flush=True)
    sys.exit(0)
for kind, dev, mode, churn in (("sort", False, "rows", 0.0), ("sort", True, "rows", 0.0), ("visual", False, "rows", 0.0), ("visual", True, "rows", 0.0),
                               ("visual", True, "pinned", 0.0), ("visual", True, "device", 0.0),
                               ("sort", True, "rows", 0.05), ("visual", True, "pinned", 0.05), ("visual", True, "device", 0.05)):
    ms, matched, rows = run(kind, dev, mode, churn)
    print(json.dumps({"tracker": kind, "objects": n, "feature_len": d if kind == "visual" else 0, "upkeep": "device" if dev else "host",
                      "features": mode if kind == "visual" else None, "churn_per_frame": churn, "ms_per_frame_median": round(ms, 3),
                      "tracks_continued_last_frame": matched, "table_rows_last_frame": rows}), flush=True)
