"""Child of tests/test_trackers.py::test_caller_may_overwrite_its_device_feature_block_once_predict_has_returned (torch first: its HIP
context wants to be the first one of the process)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

torch.zeros(1, device="cuda:0")

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from similari_amd import abi, synth  # noqa: E402
from similari_amd import trackers as TR  # noqa: E402

rng = np.random.default_rng(5)
n, d, frames = 1000, 2048, 10   # (long rows, a deep bank: the bank dispatches outlast the return of predict() by tens of microseconds)
opts = (TR.VisualSortOptions().max_idle_epochs(3).kept_history_length(2).visual_metric(TR.VisualSortMetricType.cosine(0.2))
        .positional_metric(TR.PositionalMetricType.iou(0.3)).visual_minimal_track_length(1).visual_max_observations(8).visual_min_votes(1))
a = TR.VisualSort(opts=opts, feature_len=d, device_upkeep=True)
b = TR.VisualSort(opts=opts, feature_len=d, device_upkeep=True)
fresh = torch.empty((frames, n, d), dtype=torch.float32, device="cuda:0")
reused = torch.empty((n, d), dtype=torch.float32, device="cuda:0")
junk = torch.full((n, d), 7.0, dtype=torch.float32, device="cuda:0")
lib = a.lib
assert lib.sa_device_block_register(C.c_void_p(fresh.data_ptr()), fresh.numel() * 4, 0) == 0
assert lib.sa_device_block_register(C.c_void_p(reused.data_ptr()), reused.numel() * 4, 0) == 0
ident = synth.reid_identities(rng, n, d)
world = synth.dense_boxes(rng, n, (1920.0, 1080.0))
keep = []
fp, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
placeholder = np.zeros(d, np.float32)
last = None
for f in range(frames):
    world = synth.jitter_boxes(rng, world, 2.0)
    feats = torch.from_numpy(synth.observe(rng, ident, 0.01))
    fresh[f].copy_(feats)
    reused.copy_(feats)
    torch.cuda.synchronize()
    boxes = [TR.Universal2DBox(float(r["xc"]), float(r["yc"]), None, float(r["aspect"]), float(r["height"]), float(r["confidence"])) for r in world]
    outs = []
    for trk, base in ((a, fresh.data_ptr() + f * n * d * 4), (b, reused.data_ptr())):
        items = [TR.VisualSortObservation(placeholder, 0.9, bx, None) for bx in boxes]   # (the pointers are set below)
        arr = trk._obs_array(items, keep)
        for k in range(n):
            arr[k].feature = C.cast(C.c_void_p(base + k * d * 4), fp)
        out = (abi.sa_sort_track * n)()
        assert lib.sa_tracker_predict(trk.h, 0, n, arr, out) == 0, lib.sa_tracker_last_error(trk.h)
        if trk is b:
            reused.copy_(junk)   # queued on torch's stream the moment predict() is back: nothing of the engine may still read the block
        outs.append([(o.id, o.length, o.voting_type) for o in out])
    assert outs[0] == outs[1], f"frame {f}: the tracks differ"
    last = outs[0]
torch.cuda.synchronize()
assert sum(1 for t in last if t[1] > 1) > n // 2, "the loop should continue most tracks"
checked = 0
for tid in [t[0] for t in last[:300]]:
    banks = []
    for trk in (a, b):
        eng = trk.lib.sa_tracker_engine(trk.h)
        q, pres, ft = np.zeros(8, np.float32), np.zeros(8, np.uint8), np.zeros((8, d), np.float32)
        assert trk.lib.sa_tracks_get_state(eng, 0, tid, None, None, q.ctypes.data_as(fp), pres.ctypes.data_as(u8p), ft.ctypes.data_as(fp)) == 0
        banks.append((pres, ft))
    np.testing.assert_array_equal(banks[0][0], banks[1][0])
    np.testing.assert_array_equal(banks[0][1], banks[1][1])
    checked += int(banks[0][0].sum())
assert checked > 300
a.close()
b.close()
lib.sa_device_block_unregister(C.c_void_p(fresh.data_ptr()))
lib.sa_device_block_unregister(C.c_void_p(reused.data_ptr()))
print("OVERWRITE-OK", checked)
