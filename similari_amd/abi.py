"""ctypes mirror of include/similari_assoc.h (the C ABI of the association engine).

Only POD layouts and function prototypes live here; nothing in this module computes anything.
The shared library is built in-tree by `__graft_entry__.build()` / `similari_amd.build`.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "lib" / "libsimilari_assoc.so"

SA_OK = 0
SA_ERR_BAD_ARG = -1
SA_ERR_OOM = -2
SA_ERR_HIP = -3
SA_ERR_UNSUPPORTED = -4
SA_ERR_NOT_FOUND = -5
SA_ERR_STATE = -6
SA_ERR_NO_DEVICE = -7

SA_POS_IOU = 0
SA_POS_MAHALANOBIS = 1
SA_VIS_NONE = 0
SA_VIS_COSINE = 1
SA_VIS_EUCLIDEAN = 2
SA_VOTE_NONE = 0
SA_VOTE_VISUAL = 1
SA_VOTE_POSITIONAL = 2

SA_FLAG_PROFILE = 0x2
SA_FLAG_GRAPH = 0x8
SA_FLAG_FUSED_FRAME = 0x10
SA_FLAG_SEPARATE_FRAME = 0x20
SA_FLAG_TAP = 0x80
SA_FLAG_GENERAL_TAIL = 0x100
SA_FLAG_NEVER_LEAN = 0x200
SA_FLAG_SEPARATE_RESOLVE = 0x400
SA_FLAG_EUCLID_VALU = 0x800
SA_FLAG_EUCLID_MFMA = 0x1000
SA_FLAG_BESTFIT_TILE = 0x2000
SA_FLAG_XCD_TILES = 0x4000
SA_FLAG_ROW_TILES = 0x8000
SA_FLAG_SIGNAL_COMPLETION = 0x10000
SA_FLAG_STAGED_LOOP = 0x20000
SA_FLAG_NO_YIELD = 0x40000

# Path switches OR-ed into every config make_config builds (tests: the `sa_path` fixture of tests/conftest.py sends whole parity tests
# through the engine's other paths in the same process) and a tile-plan override for the same purpose.
EXTRA_FLAGS = 0


class sa_box(C.Structure):
    _fields_ = [
        ("xc", C.c_float),
        ("yc", C.c_float),
        ("angle", C.c_float),
        ("aspect", C.c_float),
        ("height", C.c_float),
        ("confidence", C.c_float),
        ("has_angle", C.c_int32),
        ("reserved", C.c_int32),
    ]


BOX_DTYPE = np.dtype(
    [
        ("xc", "<f4"),
        ("yc", "<f4"),
        ("angle", "<f4"),
        ("aspect", "<f4"),
        ("height", "<f4"),
        ("confidence", "<f4"),
        ("has_angle", "<i4"),
        ("reserved", "<i4"),
    ]
)
assert BOX_DTYPE.itemsize == C.sizeof(sa_box) == 32


class sa_config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("stream", C.c_void_p),
        ("positional_kind", C.c_int32),
        ("positional_threshold", C.c_float),
        ("positional_min_confidence", C.c_float),
        ("visual_kind", C.c_int32),
        ("visual_threshold", C.c_float),
        ("feature_len", C.c_uint32),
        ("max_observations", C.c_uint32),
        ("visual_min_votes", C.c_uint32),
        ("visual_minimal_track_length", C.c_uint32),
        ("visual_minimal_area", C.c_float),
        ("visual_minimal_quality_use", C.c_float),
        ("visual_minimal_own_area_percentage_use", C.c_float),
        ("max_idle_epochs", C.c_uint64),
        ("n_constraints", C.c_uint32),
        ("constraint_epoch_delta", C.POINTER(C.c_uint64)),
        ("constraint_max_dist", C.POINTER(C.c_float)),
        ("kf_position_weight", C.c_float),
        ("kf_velocity_weight", C.c_float),
        ("flags", C.c_uint32),
        ("visual_minimal_quality_collect", C.c_float),
        ("visual_minimal_own_area_percentage_collect", C.c_float),
        ("gemm_plan", C.c_int32),
        ("euclid_backoff_frames", C.c_uint32),
        ("poll_spin_us", C.c_int32),
    ]


class sa_tracks(C.Structure):
    _fields_ = [
        ("n", C.c_uint32),
        ("ids", C.POINTER(C.c_uint64)),
        ("boxes", C.POINTER(sa_box)),
        ("epochs", C.POINTER(C.c_uint64)),
        ("kf_mean", C.POINTER(C.c_float)),
        ("kf_cov", C.POINTER(C.c_float)),
        ("feats", C.POINTER(C.c_float)),
        ("feat_present", C.POINTER(C.c_uint8)),
    ]


class sa_detections(C.Structure):
    _fields_ = [
        ("n", C.c_uint32),
        ("boxes", C.POINTER(sa_box)),
        ("feats", C.POINTER(C.c_float)),
        ("feat_present", C.POINTER(C.c_uint8)),
        ("feat_quality", C.POINTER(C.c_float)),
        ("own_area", C.POINTER(C.c_float)),
    ]


class sa_scene_request(C.Structure):
    _fields_ = [("scene_id", C.c_uint64), ("epoch", C.c_uint64), ("detections", sa_detections)]


class sa_scene_result(C.Structure):
    _fields_ = [("out_track_id", C.POINTER(C.c_uint64)), ("out_voting_type", C.POINTER(C.c_uint8))]


class sa_kernel_stat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


class sa_sort_track(C.Structure):
    _fields_ = [
        ("id", C.c_uint64),
        ("epoch", C.c_uint64),
        ("predicted_bbox", sa_box),
        ("observed_bbox", sa_box),
        ("scene_id", C.c_uint64),
        ("length", C.c_uint64),
        ("voting_type", C.c_int32),
        ("has_custom_object_id", C.c_int32),
        ("custom_object_id", C.c_int64),
    ]


class sa_observation(C.Structure):
    _fields_ = [
        ("bbox", sa_box),
        ("feature", C.POINTER(C.c_float)),
        ("feature_quality", C.c_float),
        ("own_area", C.c_float),
        ("has_custom_object_id", C.c_int32),
        ("reserved", C.c_int32),
        ("custom_object_id", C.c_int64),
    ]


class sa_tracker_options(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("visual", C.c_int32),
        ("batch_ids", C.c_int32),
        ("history_length", C.c_uint32),
        ("auto_waste_periodicity", C.c_uint32),
        ("max_idle_epochs", C.c_uint64),
        ("positional_kind", C.c_int32),
        ("positional_threshold", C.c_float),
        ("positional_min_confidence", C.c_float),
        ("n_constraints", C.c_uint32),
        ("constraint_epoch_delta", C.POINTER(C.c_uint64)),
        ("constraint_max_dist", C.POINTER(C.c_float)),
        ("kalman_position_weight", C.c_float),
        ("kalman_velocity_weight", C.c_float),
        ("visual_kind", C.c_int32),
        ("visual_threshold", C.c_float),
        ("feature_len", C.c_uint32),
        ("visual_max_observations", C.c_uint32),
        ("visual_min_votes", C.c_uint32),
        ("visual_minimal_track_length", C.c_uint32),
        ("visual_minimal_area", C.c_float),
        ("visual_minimal_quality_use", C.c_float),
        ("visual_minimal_quality_collect", C.c_float),
        ("visual_minimal_own_area_percentage_use", C.c_float),
        ("visual_minimal_own_area_percentage_collect", C.c_float),
        ("device_upkeep", C.c_int32),
        ("workers", C.c_int32),
        ("n_devices", C.c_uint32),
        ("devices", C.POINTER(C.c_int32)),
        ("spin_us", C.c_int32),
    ]


def _ptr(arr, ctype):
    if arr is None:
        return C.cast(None, C.POINTER(ctype))
    return arr.ctypes.data_as(C.POINTER(ctype))


def make_boxes(xc, yc, aspect, height, confidence=None, angle=None) -> np.ndarray:
    """Build an sa_box array (numpy structured, C-contiguous). angle=None -> Option::None."""
    n = len(xc)
    b = np.zeros(n, dtype=BOX_DTYPE)
    b["xc"] = np.asarray(xc, np.float32)
    b["yc"] = np.asarray(yc, np.float32)
    b["aspect"] = np.asarray(aspect, np.float32)
    b["height"] = np.asarray(height, np.float32)
    b["confidence"] = 1.0 if confidence is None else np.asarray(confidence, np.float32)
    if angle is not None:
        b["angle"] = np.asarray(angle, np.float32)
        b["has_angle"] = 1
    return b


def ltwh(l, t, w, h, confidence=1.0) -> np.ndarray:
    """BoundingBox(left, top, width, height) -> Universal2DBox (bbox.rs:246-257), f32 arithmetic."""
    l, t, w, h = (np.atleast_1d(np.asarray(v, np.float32)) for v in (l, t, w, h))
    two = np.float32(2.0)
    return make_boxes(l + w / two, t + h / two, w / h, h, confidence=np.float32(confidence))


class Keep:
    """Holds numpy arrays alive for the lifetime of a ctypes struct that points into them."""

    def __init__(self):
        self.refs = []

    def arr(self, a, dtype, shape=None):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=dtype)
        if shape is not None:
            a = a.reshape(shape)
        self.refs.append(a)
        return a


def make_config(
    positional="iou",
    positional_threshold=0.3,
    positional_min_confidence=0.05,
    visual=None,
    visual_threshold=0.0,
    feature_len=0,
    max_observations=1,
    visual_min_votes=1,
    visual_minimal_track_length=1,
    visual_minimal_area=0.0,
    visual_minimal_quality_use=0.0,
    visual_minimal_own_area_percentage_use=0.0,
    max_idle_epochs=5,
    constraints=(),
    kf_position_weight=1.0 / 20.0,
    kf_velocity_weight=1.0 / 160.0,
    device=-1,
    stream=None,
    flags=0,
    visual_minimal_quality_collect=0.0,
    visual_minimal_own_area_percentage_collect=0.0,
    gemm_plan=None,
    euclid_backoff_frames=0,
    poll_spin_us=0,
):
    """sa_config with the reference's defaults; returns (cfg, keepalive)."""
    keep = Keep()
    cfg = sa_config()
    cfg.struct_size = C.sizeof(sa_config)
    cfg.device = device
    cfg.stream = stream
    cfg.positional_kind = SA_POS_IOU if positional == "iou" else SA_POS_MAHALANOBIS
    cfg.positional_threshold = positional_threshold
    cfg.positional_min_confidence = positional_min_confidence
    cfg.visual_kind = {None: SA_VIS_NONE, "cosine": SA_VIS_COSINE, "euclidean": SA_VIS_EUCLIDEAN}[visual]
    cfg.visual_threshold = visual_threshold
    cfg.feature_len = feature_len
    cfg.max_observations = max_observations
    cfg.visual_min_votes = visual_min_votes
    cfg.visual_minimal_track_length = visual_minimal_track_length
    cfg.visual_minimal_area = visual_minimal_area
    cfg.visual_minimal_quality_use = visual_minimal_quality_use
    cfg.visual_minimal_own_area_percentage_use = visual_minimal_own_area_percentage_use
    cfg.max_idle_epochs = max_idle_epochs
    # SpatioTemporalConstraints::add_constraints sorts by delta and dedups (first wins after a stable sort)
    cons = sorted(constraints, key=lambda c: c[0])
    dedup = []
    for d, m in cons:
        if not dedup or dedup[-1][0] != d:
            dedup.append((d, m))
    cfg.n_constraints = len(dedup)
    deltas = keep.arr([d for d, _ in dedup], np.uint64) if dedup else None
    dists = keep.arr([m for _, m in dedup], np.float32) if dedup else None
    cfg.constraint_epoch_delta = _ptr(deltas, C.c_uint64)
    cfg.constraint_max_dist = _ptr(dists, C.c_float)
    cfg.kf_position_weight = kf_position_weight
    cfg.kf_velocity_weight = kf_velocity_weight
    cfg.flags = flags | EXTRA_FLAGS
    cfg.gemm_plan = 0 if gemm_plan is None else int(gemm_plan) + 1
    cfg.euclid_backoff_frames = euclid_backoff_frames
    cfg.poll_spin_us = poll_spin_us
    cfg.visual_minimal_quality_collect = visual_minimal_quality_collect
    cfg.visual_minimal_own_area_percentage_collect = visual_minimal_own_area_percentage_collect
    cfg._keep = keep
    return cfg


def make_tracks(ids, boxes, epochs, kf_mean=None, kf_cov=None, feats=None, feat_present=None):
    keep = Keep()
    t = sa_tracks()
    ids = keep.arr(ids, np.uint64)
    t.n = len(ids)
    boxes = keep.arr(boxes, BOX_DTYPE)
    t.ids = _ptr(ids, C.c_uint64)
    t.boxes = C.cast(boxes.ctypes.data, C.POINTER(sa_box)) if t.n else C.cast(None, C.POINTER(sa_box))
    t.epochs = _ptr(keep.arr(epochs, np.uint64), C.c_uint64)
    t.kf_mean = _ptr(keep.arr(kf_mean, np.float32), C.c_float)
    t.kf_cov = _ptr(keep.arr(kf_cov, np.float32), C.c_float)
    t.feats = _ptr(keep.arr(feats, np.float32), C.c_float)
    t.feat_present = _ptr(keep.arr(feat_present, np.uint8), C.c_uint8)
    t._keep = keep
    return t


def make_detections(boxes, feats=None, feat_present=None, feat_quality=None, own_area=None, feats_device_ptr=None):
    """feats: N x D float32 in host memory, or feats_device_ptr: the address of such rows inside a block registered with
    sa_device_block_register (e.g. tensor.data_ptr() of a CUDA/HIP tensor the caller keeps alive and does not write to meanwhile)."""
    keep = Keep()
    d = sa_detections()
    boxes = keep.arr(boxes, BOX_DTYPE)
    d.n = len(boxes)
    d.boxes = C.cast(boxes.ctypes.data, C.POINTER(sa_box)) if d.n else C.cast(None, C.POINTER(sa_box))
    if feats_device_ptr is not None:
        assert feats is None
        d.feats = C.cast(C.c_void_p(int(feats_device_ptr)), C.POINTER(C.c_float))
    else:
        d.feats = _ptr(keep.arr(feats, np.float32), C.c_float)
    d.feat_present = _ptr(keep.arr(feat_present, np.uint8), C.c_uint8)
    d.feat_quality = _ptr(keep.arr(feat_quality, np.float32), C.c_float)
    d.own_area = _ptr(keep.arr(own_area, np.float32), C.c_float)
    d._keep = keep
    return d


# ---- prototypes of every symbol include/similari_assoc.h declares -------------------------------
u32, u64, i32, f64p = C.c_uint32, C.c_uint64, C.c_int32, C.POINTER(C.c_double)
P = C.POINTER
ENGINE = C.c_void_p
PROTOTYPES = {
    "sa_config_default": (None, [P(sa_config)]),
    "sa_engine_create": (C.c_int, [P(sa_config), P(ENGINE)]),
    "sa_engine_destroy": (None, [ENGINE]),
    "sa_last_error": (C.c_char_p, [ENGINE]),
    "sa_api_version": (u32, []),
    "sa_tracks_upsert": (C.c_int, [ENGINE, u64, P(sa_tracks)]),
    "sa_tracks_remove": (C.c_int, [ENGINE, u64, u32, P(u64)]),
    "sa_tracks_remove_many": (C.c_int, [ENGINE, u32, P(u64), P(u32), P(P(u64))]),
    "sa_tracks_remove_stage": (C.c_int, [ENGINE, u64, u32, P(u64)]),
    "sa_tracks_remove_commit": (C.c_int, [ENGINE]),
    "sa_tracks_remove_abort": (C.c_int, [ENGINE]),
    "sa_tracks_count": (C.c_int, [ENGINE, u64, P(u32)]),
    "sa_tracks_order": (C.c_int, [ENGINE, u64, P(u64), u32, P(u32)]),
    "sa_associate": (C.c_int, [ENGINE, u64, u64, P(sa_detections), P(u64), P(C.c_uint8)]),
    "sa_batch_begin": (C.c_int, [ENGINE]),
    "sa_batch_add": (C.c_int, [ENGINE, u64, u64, P(sa_detections), P(u32)]),
    "sa_batch_add_rows": (C.c_int, [ENGINE, u64, u64, P(sa_detections), P(P(C.c_float)), P(u32)]),
    "sa_batch_add_deferred": (C.c_int, [ENGINE, u64, u64, P(sa_detections), P(P(C.c_float)), P(u32)]),
    "sa_batch_fill": (C.c_int, [ENGINE, u32]),
    "sa_batch_run": (C.c_int, [ENGINE]),
    "sa_batch_sync": (C.c_int, [ENGINE]),
    "sa_batch_fetch": (C.c_int, [ENGINE, u32, P(u64), P(C.c_uint8)]),
    "sa_batch_fetch_cols": (C.c_int, [ENGINE, u32, P(i32)]),
    "sa_batch_results": (C.c_int, [ENGINE, u32, P(P(u64)), P(P(C.c_uint8)), P(P(i32))]),
    "sa_associate_batch": (C.c_int, [ENGINE, u32, P(sa_scene_request), P(sa_scene_result)]),
    "sa_pipe_stage": (C.c_int, [ENGINE, u32, P(sa_scene_request), P(u64)]),
    "sa_pipe_launch": (C.c_int, [ENGINE, u64]),
    "sa_pipe_submit": (C.c_int, [ENGINE, u32, P(sa_scene_request), P(u64)]),
    "sa_pipe_wait": (C.c_int, [ENGINE, u64, P(sa_scene_result)]),
    "sa_cluster_create": (C.c_int, [P(sa_config), u32, P(i32), P(C.c_void_p)]),
    "sa_cluster_destroy": (None, [C.c_void_p]),
    "sa_cluster_last_error": (C.c_char_p, [C.c_void_p]),
    "sa_cluster_size": (u32, [C.c_void_p]),
    "sa_cluster_shard_of": (u32, [C.c_void_p, u64]),
    "sa_cluster_engine": (C.c_void_p, [C.c_void_p, u32]),
    "sa_cluster_tracks_upsert": (C.c_int, [C.c_void_p, u64, P(sa_tracks)]),
    "sa_cluster_tracks_remove": (C.c_int, [C.c_void_p, u64, u32, P(u64)]),
    "sa_cluster_associate_batch": (C.c_int, [C.c_void_p, u32, P(sa_scene_request), P(sa_scene_result)]),
    "sa_cluster_last_ms": (C.c_double, [C.c_void_p, u32]),
    "sa_tracks_apply": (C.c_int, [ENGINE, u32, P(u64), P(sa_box)]),
    "sa_tracks_apply_begin": (C.c_int, [ENGINE, u32, P(u64)]),
    "sa_tracks_apply_end": (C.c_int, [ENGINE, u32, P(sa_box)]),
    "sa_batch_run_apply": (C.c_int, [ENGINE, P(u64), C.c_int]),
    "sa_tracks_apply_collect": (C.c_int, [ENGINE, u32, P(u64), P(sa_box)]),
    "sa_tracks_apply_collect_begin": (C.c_int, [ENGINE]),
    "sa_tracks_apply_collect_slot": (C.c_int, [ENGINE, u32, P(u64), P(sa_box)]),
    "sa_tracks_apply_collect_table": (C.c_int, [ENGINE, u32, P(u64)]),
    "sa_tracks_apply_collect_end": (C.c_int, [ENGINE]),
    "sa_tracks_get_state": (C.c_int, [ENGINE, u64, u64, P(C.c_float), P(C.c_float), P(C.c_float), P(C.c_uint8), P(C.c_float)]),
    "sa_tracks_set_state": (C.c_int, [ENGINE, u64, u64, P(C.c_float), P(C.c_float), P(C.c_float)]),
    "sa_nms": (C.c_int, [ENGINE, u32, P(sa_box), P(C.c_float), C.c_float, C.c_float, P(u32), P(u32)]),
    "sa_own_areas": (C.c_int, [ENGINE, u32, P(sa_box), P(C.c_float)]),
    "sa_host_alloc": (C.c_void_p, [C.c_uint64]),
    "sa_host_free": (None, [C.c_void_p]),
    "sa_device_block_register": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int]),
    "sa_device_block_unregister": (None, [C.c_void_p]),
    "sa_tap_dims": (C.c_int, [ENGINE, u32, P(u32), P(u32), P(u32)]),
    "sa_tap_positional": (C.c_int, [ENGINE, u32, P(C.c_float)]),
    "sa_tap_visual": (C.c_int, [ENGINE, u32, P(C.c_float)]),
    "sa_tap_quantised": (C.c_int, [ENGINE, u32, P(C.c_int64)]),
    "sa_tap_track_polygons": (C.c_int, [ENGINE, u64, f64p, u32, P(u32)]),
    "sa_tap_votes": (C.c_int, [ENGINE, u32, f64p, P(i32), f64p, P(i32), P(i32)]),
    "sa_tap_edges": (C.c_int, [ENGINE, u32, P(u32), u32, P(u32), P(C.c_int64), P(u32)]),
    "sa_profile_enable": (C.c_int, [ENGINE, C.c_int]),
    "sa_profile_reset": (C.c_int, [ENGINE]),
    "sa_profile_read": (C.c_int, [ENGINE, P(sa_kernel_stat), u32, P(u32)]),
    "sa_batch_time": (C.c_int, [ENGINE, u32, f64p]),
    "sa_feature_distance_matrix": (
        C.c_int,
        [ENGINE, i32, u32, u32, u32, P(C.c_float), P(C.c_float), P(C.c_float), u32, f64p],
    ),
    # include/similari_tracker.h
    "sa_tracker_options_default": (None, [P(sa_tracker_options), C.c_int]),
    "sa_tracker_create": (C.c_int, [P(sa_tracker_options), P(C.c_void_p)]),
    "sa_tracker_destroy": (None, [C.c_void_p]),
    "sa_tracker_last_error": (C.c_char_p, [C.c_void_p]),
    "sa_tracker_predict": (C.c_int, [C.c_void_p, u64, u32, P(sa_observation), P(sa_sort_track)]),
    "sa_tracker_predict_batch": (C.c_int, [C.c_void_p, u32, P(u64), P(u32), P(P(sa_observation)), P(P(sa_sort_track))]),
    "sa_tracker_predict_batch_begin": (C.c_int, [C.c_void_p, u32, P(u64), P(u32), P(P(sa_observation)), P(C.c_void_p)]),
    "sa_batch_result_size": (u32, [C.c_void_p]),
    "sa_batch_result_ready": (C.c_int, [C.c_void_p]),
    "sa_batch_result_get": (C.c_int, [C.c_void_p, P(u64), P(sa_sort_track), u32, P(u32)]),
    "sa_batch_result_take": (C.c_int, [C.c_void_p, P(u64), P(P(sa_sort_track)), P(u32)]),
    "sa_batch_result_free": (None, [C.c_void_p]),
    "sa_tracker_idle_tracks": (C.c_int, [C.c_void_p, u64, P(sa_sort_track), u32, P(u32)]),
    "sa_tracker_skip_epochs": (C.c_int, [C.c_void_p, u64, u64]),
    "sa_tracker_current_epoch": (C.c_int, [C.c_void_p, u64, P(u64)]),
    "sa_tracker_wasted": (C.c_int, [C.c_void_p, P(sa_sort_track), u32, P(u32)]),
    "sa_tracker_clear_wasted": (C.c_int, [C.c_void_p]),
    "sa_tracker_active_tracks": (C.c_int, [C.c_void_p, P(u64)]),
    "sa_tracker_track_state": (C.c_int, [C.c_void_p, u64, P(C.c_float), P(C.c_float)]),
    "sa_tracker_track_info": (C.c_int, [C.c_void_p, u64, P(u64)]),
    "sa_tracker_engine": (C.c_void_p, [C.c_void_p]),
}


def hip_runtimes_mapped() -> list[str]:
    """Paths of the libamdhip64 images mapped into this process (one is healthy; two runtimes in one process do not see each other's
    devices, streams or pointers)."""
    seen = []
    try:
        with open("/proc/self/maps") as f:
            for ln in f:
                path = ln.rsplit(" ", 1)[-1].strip()
                if "libamdhip64" in path and path not in seen:
                    seen.append(path)
    except OSError:
        pass
    return seen


def _elf_dynamic_strings(path, tags=(1, 14)):
    """DT_NEEDED (1) / DT_SONAME (14) strings of a 64-bit little-endian ELF file: {tag: [strings]} ({} when it cannot be read).  Reads the
    headers, the dynamic section and the string table only (a HIP runtime is tens of megabytes)."""
    import struct

    out = {t: [] for t in tags}
    try:
        with open(path, "rb") as f:
            hdr = f.read(64)
            if hdr[:4] != b"\x7fELF" or hdr[4] != 2 or hdr[5] != 1:
                return {}
            shoff, = struct.unpack_from("<Q", hdr, 0x28)
            shentsize, shnum = struct.unpack_from("<HH", hdr, 0x3A)
            f.seek(shoff)
            sh = f.read(shentsize * shnum)
            secs = [struct.unpack_from("<IIQQQQIIQQ", sh, i * shentsize) for i in range(shnum)]
            for sec in secs:
                if sec[1] != 6:   # SHT_DYNAMIC
                    continue
                strtab = secs[sec[6]]
                f.seek(strtab[4])
                strs = f.read(strtab[5])
                f.seek(sec[4])
                dyn = f.read(sec[5])
                for off in range(0, len(dyn) - 15, 16):
                    tag, val = struct.unpack_from("<qQ", dyn, off)
                    if tag in out and val < len(strs):
                        out[tag].append(strs[val:strs.index(b"\0", val)].decode())
    except Exception:
        return {}
    return out


def share_torchs_hip_runtime(lib_path=None) -> str | None:
    """ONE HIP runtime per process.  A PyTorch-ROCm wheel ships a libamdhip64 of its own and its libraries ask for it by FILE name
    ("libamdhip64.so"), this library asks for the SONAME ("libamdhip64.so.7"): with torch loaded first the loader hands torch's copy to
    both; with this library loaded first it takes /opt/rocm's and torch then loads its own beside it — and the second runtime to start
    finds the devices taken ("No HIP GPUs are available"), let alone shares a registered feature buffer or a stream with the first.  So,
    before libsimilari_assoc.so is loaded and when no HIP runtime is mapped yet, the copy a torch installed in this interpreter would
    bring is loaded up front (found through the import machinery, torch itself is NOT imported) — either order then runs on one
    runtime.  SA_HIP_RUNTIME=system keeps the system's runtime (a process that never starts torch's GPU side)."""
    if os.environ.get("SA_HIP_RUNTIME", "") == "system" or hip_runtimes_mapped():
        return None
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    cand = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if not cand.exists():
        return None
    # Only a runtime of the SAME SONAME as the one libsimilari_assoc.so was linked against may stand in for it: a wheel built against
    # another ROCm major would have this library's gfx950 code objects registered with a runtime it was not built for (SA_HIP_RUNTIME=torch
    # forces the wheel's copy all the same).
    if os.environ.get("SA_HIP_RUNTIME", "") != "torch" and lib_path is not None:
        need = [n for n in _elf_dynamic_strings(lib_path).get(1, []) if n.startswith("libamdhip64.so")]
        have = _elf_dynamic_strings(cand).get(14, [])
        if need and have and need[0] != have[0]:
            import warnings

            warnings.warn(f"similari_amd: torch ships {have[0]} but libsimilari_assoc.so needs {need[0]}: the system's HIP runtime is used; "
                          "torch's GPU side cannot be started in this process afterwards (SA_HIP_RUNTIME=torch to share torch's copy)")
            return None
    C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
    return str(cand)


def load_library(path: os.PathLike | None = None) -> C.CDLL:
    """Load libsimilari_assoc.so and attach prototypes. Fails loudly when it is not built."""
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise RuntimeError(
            f"{p} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback."
        )
    share_torchs_hip_runtime(p)
    lib = C.CDLL(str(p))
    if len(hip_runtimes_mapped()) > 1:
        import warnings

        warnings.warn("similari_amd: more than one HIP runtime is mapped into this process (" + ", ".join(sorted(hip_runtimes_mapped())) +
                      "): the one that starts second will not see the GPUs")
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library drift
        fn.restype = res
        fn.argtypes = args
    return lib
