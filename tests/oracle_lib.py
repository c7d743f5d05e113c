"""ctypes binding of oracle/liboracle.so (CPU restatement of the reference). Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from similari_amd import abi

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
u32, u64, f32 = C.c_uint32, C.c_uint64, C.c_float
P = C.POINTER


class or_frame_out(C.Structure):
    _fields_ = [
        ("positional", P(f32)),
        ("visual", P(f32)),
        ("quantised", P(C.c_int64)),
        ("compatible", P(C.c_uint8)),
        ("track_id", P(u64)),
        ("voting_type", P(C.c_uint8)),
        ("total_weight", C.c_int64),
        ("n_distances", u64),
    ]


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", str(ORACLE_DIR)], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    so = ORACLE_DIR / "liboracle.so"
    srcs = [ORACLE_DIR / "oracle.cpp", ORACLE_DIR / "oracle_tracker.cpp", ORACLE_DIR / "oracle.h"]
    if not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        build()
    L = C.CDLL(str(so))
    B = P(abi.sa_box)
    fp = P(f32)
    dp = P(C.c_double)
    sig = {
        "or_feature_blocks": (u32, [u32]),
        "or_feature_pad": (u32, [fp, u32, fp]),
        "or_euclidean": (f32, [fp, u32, fp, u32]),
        "or_cosine": (f32, [fp, u32, fp, u32]),
        "or_radius": (f32, [B]),
        "or_area": (f32, [B]),
        "or_too_far": (C.c_int, [B, B]),
        "or_dist_in_2r": (f32, [B, B]),
        "or_vertices": (None, [B, dp]),
        "or_sh_clip": (u32, [dp, u32, dp, u32, dp]),
        "or_polygon_area": (C.c_double, [dp, u32]),
        "or_intersection": (C.c_double, [B, B]),
        "or_iou": (C.c_int, [B, B, fp]),
        "or_kf_initiate": (None, [f32, f32, B, fp, fp]),
        "or_kf_predict": (None, [f32, f32, fp, fp, fp, fp]),
        "or_kf_update": (None, [f32, f32, fp, fp, B, fp, fp]),
        "or_kf_distance": (f32, [f32, f32, fp, fp, B]),
        "or_kf_distance5": (f32, [f32, fp, fp, B]),
        "or_kf_cost": (f32, [f32, C.c_int]),
        "or_kf_state_box": (None, [fp, B]),
        "or_make_prediction": (None, [f32, f32, C.c_int, fp, fp, B, B]),
        "or_constraints_validate": (C.c_int, [u32, P(u64), fp, u64, f32]),
        "or_compatible": (C.c_int, [P(abi.sa_config), B, u64, B, u64]),
        "or_positional_metric": (C.c_int, [P(abi.sa_config), B, B, fp, fp, fp]),
        "or_kuhn_munkres": (C.c_int, [u32, u32, P(C.c_int64), P(C.c_int64), P(u32)]),
        "or_quantise": (C.c_int64, [f32]),
        "or_sort_voting": (C.c_int, [f32, u32, u32, u32, P(u64), P(u64), fp, u32, P(u64), P(u64), P(C.c_int64)]),
        "or_bestfit_voting": (C.c_int, [f32, u32, u32, P(u64), P(u64), fp, u32, P(u64), P(u64), dp]),
        "or_visual_voting": (
            C.c_int,
            [f32, f32, u32, u32, P(u64), P(u64), fp, fp, u32, P(u64), P(u64), P(C.c_uint8)],
        ),
        "or_nms": (C.c_int, [u32, B, fp, f32, f32, P(u32), P(u32)]),
        "or_own_area_shares": (C.c_int, [u32, B, fp]),
        "or_associate": (C.c_int, [P(abi.sa_config), u32, P(abi.sa_tracks), u64, P(abi.sa_detections), P(or_frame_out)]),
        "or_associate_sharded": (C.c_int, [P(abi.sa_config), u32, P(abi.sa_tracks), u64, P(abi.sa_detections), P(or_frame_out), u32]),
    }
    T = C.c_void_p
    OBS, TRK = P(abi.sa_observation), P(abi.sa_sort_track)
    sig.update({
        "or_tracker_create": (T, [P(abi.sa_tracker_options)]),
        "or_tracker_destroy": (None, [T]),
        "or_tracker_predict_batch": (C.c_int, [T, u32, P(u64), P(u32), P(OBS), P(TRK)]),
        "or_tracker_skip_epochs": (C.c_int, [T, u64, u64]),
        "or_tracker_current_epoch": (u64, [T, u64]),
        "or_tracker_active_tracks": (u64, [T]),
        "or_tracker_wasted": (u32, [T, TRK, u32]),
        "or_tracker_idle_tracks": (u32, [T, u64, TRK, u32]),
        "or_tracker_track_state": (C.c_int, [T, u64, fp, fp]),
        "or_tracker_track_info": (C.c_int, [T, u64, P(u64)]),
    })
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def box_ptr(boxes: np.ndarray, i: int = 0):
    return C.cast(boxes.ctypes.data + i * abi.BOX_DTYPE.itemsize, P(abi.sa_box))


def fptr(a):
    return a.ctypes.data_as(P(f32))


def dptr(a):
    return a.ctypes.data_as(P(C.c_double))


def nms(boxes, scores=None, nms_threshold=0.5, score_threshold=None):
    """Oracle NMS (oracle.cpp: or_nms, following src/utils/nms.rs:32-72)."""
    L = lib()
    boxes = np.ascontiguousarray(boxes, abi.BOX_DTYPE)
    keep = np.zeros(max(len(boxes), 1), np.uint32)
    m = u32()
    sc = None if scores is None else np.ascontiguousarray(scores, np.float32)
    L.or_nms(len(boxes), box_ptr(boxes), None if sc is None else fptr(sc), nms_threshold,
             float("nan") if score_threshold is None else score_threshold, keep.ctypes.data_as(P(u32)), C.byref(m))
    return keep[: m.value].copy()


def own_area_shares(boxes):
    """Oracle own-area shares (oracle.cpp: or_own_area_shares, following src/utils/clipping/bbox_own_areas.rs:8-46)."""
    boxes = np.ascontiguousarray(boxes, abi.BOX_DTYPE)
    out = np.zeros(max(len(boxes), 1), np.float32)
    lib().or_own_area_shares(len(boxes), box_ptr(boxes), fptr(out))
    return out[: len(boxes)].copy()


def associate(cfg, tracks, epoch, det, total_tracks=None, want_matrices=True, shards=1):
    """Run the oracle on one scene-frame; returns dict of numpy arrays.  shards > 1: the distance stage on that many host threads
    partitioned like the reference's TrackStore (track id % shards), one vote after them — same results."""
    L = lib()
    N, T = det.n, tracks.n
    K = max(1, cfg.max_observations) if cfg.visual_kind != abi.SA_VIS_NONE else 1
    out = or_frame_out()
    res = {
        "track_id": np.zeros(N, np.uint64),
        "voting_type": np.zeros(N, np.uint8),
    }
    if want_matrices:
        res["positional"] = np.empty((N, T), np.float32)
        res["visual"] = np.empty((N, T, K), np.float32)
        res["quantised"] = np.empty((N, T), np.int64)
        res["compatible"] = np.empty((N, T), np.uint8)
        out.positional = fptr(res["positional"])
        out.visual = fptr(res["visual"])
        out.quantised = res["quantised"].ctypes.data_as(P(C.c_int64))
        out.compatible = res["compatible"].ctypes.data_as(P(C.c_uint8))
    out.track_id = res["track_id"].ctypes.data_as(P(u64))
    out.voting_type = res["voting_type"].ctypes.data_as(P(C.c_uint8))
    rc = L.or_associate_sharded(C.byref(cfg), T if total_tracks is None else total_tracks, C.byref(tracks), epoch, C.byref(det), C.byref(out),
                                max(1, int(shards)))
    assert rc == 0
    res["total_weight"] = int(out.total_weight)
    res["n_distances"] = int(out.n_distances)
    return res


class OracleTracker:
    """The oracle's Sort / VisualSort / Batch* loops behind the same Python surface as similari_amd.trackers._Tracker."""

    def __init__(self, opts, keep=None):
        from similari_amd import trackers as TR

        self.TR = TR
        self.L = lib()
        self.opts = opts
        self._keep = keep
        self.h = self.L.or_tracker_create(C.byref(opts))

    def close(self):
        if self.h:
            self.L.or_tracker_destroy(self.h)
            self.h = None

    _obs_array = None

    def predict_batch_raw(self, scenes: dict):
        TR = self.TR
        keys = list(scenes.keys())
        keep = []
        arrs = [TR._Tracker._obs_array(self, scenes[s], keep) for s in keys]
        outs = [(abi.sa_sort_track * max(1, len(scenes[s])))() for s in keys]
        ids = (C.c_uint64 * max(1, len(keys)))(*keys)
        counts = (C.c_uint32 * max(1, len(keys)))(*[len(scenes[s]) for s in keys])
        pa = (P(abi.sa_observation) * max(1, len(keys)))(*[C.cast(a, P(abi.sa_observation)) for a in arrs])
        po = (P(abi.sa_sort_track) * max(1, len(keys)))(*[C.cast(o, P(abi.sa_sort_track)) for o in outs])
        rc = self.L.or_tracker_predict_batch(self.h, len(keys), ids, counts, pa, po)
        assert rc == 0
        return {s: [TR.SortTrack.from_c(outs[k][i]) for i in range(len(scenes[s]))] for k, s in enumerate(keys)}

    def predict_with_scene(self, scene_id, items):
        return self.predict_batch_raw({scene_id: list(items)})[scene_id]

    def predict(self, items):
        return self.predict_with_scene(0, items)

    def predict_batch(self, batch):
        return self.predict_batch_raw(batch.scenes)

    def skip_epochs_for_scene(self, scene_id, n):
        self.L.or_tracker_skip_epochs(self.h, scene_id, n)

    def current_epoch_with_scene(self, scene_id):
        return self.L.or_tracker_current_epoch(self.h, scene_id)

    def current_epoch(self):
        return self.current_epoch_with_scene(0)

    def active_tracks(self):
        return self.L.or_tracker_active_tracks(self.h)

    def wasted(self):
        n = self.L.or_tracker_wasted(self.h, None, 0)
        out = (abi.sa_sort_track * max(1, n))()
        n = self.L.or_tracker_wasted(self.h, out, n)
        return [self.TR.SortTrack.from_c(out[i]) for i in range(n)]

    def wasted_count(self):
        return self.L.or_tracker_wasted(self.h, None, 0)

    def clear_wasted(self):
        n = self.L.or_tracker_wasted(self.h, None, 0)
        out = (abi.sa_sort_track * max(1, n))()
        self.L.or_tracker_wasted(self.h, out, n)

    def idle_tracks_with_scene(self, scene_id):
        n = self.L.or_tracker_idle_tracks(self.h, scene_id, None, 0)
        out = (abi.sa_sort_track * max(1, n))()
        n = self.L.or_tracker_idle_tracks(self.h, scene_id, out, n)
        return [self.TR.SortTrack.from_c(out[i]) for i in range(n)]

    def track_state(self, track_id):
        m = np.zeros(10, np.float32)
        c = np.zeros(100, np.float32)
        assert self.L.or_tracker_track_state(self.h, track_id, fptr(m), fptr(c)) == 0
        return m, c

    def track_info(self, track_id):
        out = (C.c_uint64 * 4)()
        assert self.L.or_tracker_track_info(self.h, track_id, out) == 0
        return dict(visual_features_collected_count=out[0], observations=out[1], history=out[2], track_length=out[3])
