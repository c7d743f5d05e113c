"""BatchSort / BatchVisualSort::predict through the facade (device upkeep): S scenes of n objects per call, the C arrays built before the
timed loop and the C calls back to back (see scripts/bench_tracker.py).  BASELINE C3's shape by default: 8 scenes x 500 objects.
`async`: the call goes through sa_tracker_predict_batch_begin and the scenes are taken from the result handle (time until the call
returns / until the first scene / until the last).  `device`: the scenes' feature rows lie in one registered device block (a ReID
model's output buffer).  `device_association_us`: the association kernels of the last request set alone, replayed on resident inputs
(sa_batch_time); the device-side span of a whole predict() — association + upkeep dispatches — is what scripts/batch_tracker_timeline.sh
reads from a rocprofv3 kernel trace of this script.
`devices`: a comma-separated list of HIP ordinals — the tracker becomes a device GROUP (sa_tracker_options.devices: one engine per entry,
scenes dealt out scene_id % n; "0,0" = two engines on one GPU); `spin_us`: sa_tracker_options.spin_us (-1 = defaults, 0 = no idle
polling).  `cpu_seconds_per_1000_predicts`: user + system CPU time of the WHOLE process (caller, pool, drivers, runtime threads) over the
timed loop, scaled — what a host pays for the back-to-back loop, idle spinning included.
   python scripts/bench_batch_tracker.py [sort|visual] [scenes] [objects] [feature_len] [frames] [workers] [sync|async] [rows|device] [devices|-] [spin_us]"""
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
workers = int(sys.argv[6]) if len(sys.argv) > 6 else 0
mode = sys.argv[7] if len(sys.argv) > 7 else "sync"
feats_mode = sys.argv[8] if len(sys.argv) > 8 else "rows"
devices = [int(x) for x in sys.argv[9].split(",")] if len(sys.argv) > 9 and sys.argv[9] != "-" else None
spin_us = int(sys.argv[10]) if len(sys.argv) > 10 else -1
K = 3
rng = np.random.default_rng(0)
dev = None
if kind == "visual" and feats_mode == "device":
    import torch

    torch.zeros(1, device="cuda:0")   # torch's HIP context first (it stands in for the ReID model that owns the feature buffers)
    dev = torch.empty((frames, S, n, d), dtype=torch.float32, device="cuda:0")
if kind == "visual":
    opts = (TR.VisualSortOptions().max_idle_epochs(3).kept_history_length(3).visual_metric(TR.VisualSortMetricType.cosine(0.2))
            .positional_metric(TR.PositionalMetricType.iou(0.3)).visual_minimal_track_length(1).visual_max_observations(K))
    trk = TR.BatchVisualSort(voting_shards=workers, opts=opts, feature_len=d, device_upkeep=True, devices=devices, spin_us=spin_us)
else:
    trk = TR.BatchSort(voting_shards=workers, bbox_history=3, max_idle_epochs=3, device_upkeep=True, devices=devices, spin_us=spin_us)
lib = trk.lib
if dev is not None:
    lib.sa_device_block_register(C.c_void_p(dev.data_ptr()), dev.numel() * 4, 0)
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
        arr = trk._obs_array(items, keep)
        if dev is not None:
            dev[f, s].copy_(torch.from_numpy(feats))
            base = dev.data_ptr() + ((f * S + s) * n) * d * 4
            for k in range(n):
                arr[k].feature = C.cast(C.c_void_p(base + k * d * 4), C.POINTER(C.c_float))
        arrs.append(arr)
    outs = [(abi.sa_sort_track * n)() for _ in range(S)]
    ids = (C.c_uint64 * S)(*range(S))
    counts = (C.c_uint32 * S)(*([n] * S))
    pa = (C.POINTER(abi.sa_observation) * S)(*[C.cast(a, C.POINTER(abi.sa_observation)) for a in arrs])
    po = (C.POINTER(abi.sa_sort_track) * S)(*[C.cast(o, C.POINTER(abi.sa_sort_track)) for o in outs])
    calls.append((arrs, outs, ids, counts, pa, po))
if dev is not None:
    torch.cuda.synchronize()
import resource
times, t_begin, t_first = [], [], []
cpu0 = wall0 = None
sid, cnt = C.c_uint64(), C.c_uint32()
view = C.POINTER(abi.sa_sort_track)()
for ci, (arrs, outs, ids, counts, pa, po) in enumerate(calls):
    if ci == 3:
        ru = resource.getrusage(resource.RUSAGE_SELF)
        cpu0, wall0 = ru.ru_utime + ru.ru_stime, time.perf_counter()
    t0 = time.perf_counter()
    if mode == "async":
        h = C.c_void_p()
        rc = lib.sa_tracker_predict_batch_begin(trk.h, S, ids, counts, pa, C.byref(h))
        t1 = time.perf_counter()
        assert rc == 0, lib.sa_tracker_last_error(trk.h)
        for k in range(S):
            rc = lib.sa_batch_result_take(h, C.byref(sid), C.byref(view), C.byref(cnt))   # (in place: the reference's get() moves the scene's Vec)
            if k == 0:
                t2 = time.perf_counter()
            assert rc == 0
        times.append(time.perf_counter() - t0)
        t_begin.append(t1 - t0)
        t_first.append(t2 - t0)
        lib.sa_batch_result_free(h)
    else:
        rc = lib.sa_tracker_predict_batch(trk.h, S, ids, counts, pa, po)
        times.append(time.perf_counter() - t0)
        assert rc == 0, lib.sa_tracker_last_error(trk.h)
ru = resource.getrusage(resource.RUSAGE_SELF)
cpu_s, wall_s = ru.ru_utime + ru.ru_stime - cpu0, time.perf_counter() - wall0
cont = sum(1 for o in calls[-1][1] for i in range(n) if o[i].length > 1) if mode != "async" else None
# the device's share: the association of the last request set against the tables as they stand (resident inputs, kernels only)
eng = lib.sa_tracker_engine(trk.h)
lib.sa_batch_sync(eng)
ms = C.c_double()
dev_assoc_us = None
lib.sa_batch_time(eng, 2, C.byref(ms))   # (the first replay behind a loop re-stages the set on the host: ~9 ms at 64 scenes, no kernel in it)
if devices is None and lib.sa_batch_time(eng, 20, C.byref(ms)) == 0:
    dev_assoc_us = round(1e3 * ms.value / 20, 1)
trk.close()
if dev is not None:
    lib.sa_device_block_unregister(C.c_void_p(dev.data_ptr()))
med = float(np.median(times[3:]))
out = {"tracker": "Batch" + ("VisualSort" if kind == "visual" else "Sort"), "scenes": S, "objects_per_scene": n, "feature_len": d if kind == "visual" else 0,
       "bank": K if kind == "visual" else 0, "features": feats_mode if kind == "visual" else None, "upkeep": "device", "workers": workers, "call": mode,
       "us_per_predict_median": round(1e6 * med, 1), "us_per_predict_min": round(1e6 * float(np.min(times[3:])), 1),
       "us_per_scene": round(1e6 * med / S, 1), "device_association_us": dev_assoc_us, "tracks_continued_last_frame": cont}
out["devices"] = devices
out["spin_us"] = spin_us
out["cpu_seconds_per_1000_predicts"] = round(1e3 * cpu_s / max(1, len(calls) - 3), 4)
out["cpu_cores_busy_over_the_loop"] = round(cpu_s / wall_s, 2)
if mode == "async":
    out["us_until_begin_returns"] = round(1e6 * float(np.median(t_begin[3:])), 1)
    out["us_until_first_scene"] = round(1e6 * float(np.median(t_first[3:])), 1)
print(json.dumps(out))
