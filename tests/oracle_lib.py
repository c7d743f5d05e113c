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
        "or_associate": (C.c_int, [P(abi.sa_config), u32, P(abi.sa_tracks), u64, P(abi.sa_detections), P(or_frame_out)]),
    }
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


def associate(cfg, tracks, epoch, det, total_tracks=None, want_matrices=True):
    """Run the oracle on one scene-frame; returns dict of numpy arrays."""
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
    rc = L.or_associate(C.byref(cfg), T if total_tracks is None else total_tracks, C.byref(tracks), epoch, C.byref(det), C.byref(out))
    assert rc == 0
    res["total_weight"] = int(out.total_weight)
    res["n_distances"] = int(out.n_distances)
    return res
