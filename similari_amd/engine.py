"""Thin Python host binding of the C ABI (include/similari_assoc.h).  No arithmetic happens here: every call
lands in libsimilari_assoc.so, whose compute path is HIP-only (no CPU fallback)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"similari_assoc error {code}: {msg}")
        self.code = code


class Engine:
    """One association engine = one GPU stream set. Mirrors sa_engine_* 1:1."""

    def __init__(self, cfg: abi.sa_config, lib: C.CDLL | None = None):
        self.lib = lib or abi.load_library()
        self.cfg = cfg
        self.h = abi.ENGINE()
        rc = self.lib.sa_engine_create(C.byref(cfg), C.byref(self.h))
        if rc != abi.SA_OK:
            msg = self.lib.sa_last_error(None)
            raise EngineError(rc, msg.decode() if msg else "")
        self.K = cfg.max_observations if cfg.visual_kind != abi.SA_VIS_NONE else 1

    @classmethod
    def borrowed(cls, lib, handle, cfg=None):
        """Non-owning view of an engine that belongs to someone else (a tracker facade: sa_tracker_engine)."""
        self = cls.__new__(cls)
        self.lib, self.cfg, self.h, self._borrowed = lib, cfg, abi.ENGINE(handle), True
        self.K = 1
        return self

    def close(self):
        if self.h and not getattr(self, "_borrowed", False):
            self.lib.sa_engine_destroy(self.h)
        self.h = abi.ENGINE()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != abi.SA_OK:
            msg = self.lib.sa_last_error(self.h)
            raise EngineError(rc, msg.decode() if msg else "")

    # ---- track state ----
    def upsert(self, scene: int, tracks: abi.sa_tracks):
        self._chk(self.lib.sa_tracks_upsert(self.h, scene, C.byref(tracks)))

    def remove(self, scene: int, ids):
        ids = np.ascontiguousarray(ids, np.uint64)
        self._chk(self.lib.sa_tracks_remove(self.h, scene, len(ids), ids.ctypes.data_as(C.POINTER(C.c_uint64))))

    def count(self, scene: int) -> int:
        n = C.c_uint32()
        self._chk(self.lib.sa_tracks_count(self.h, scene, C.byref(n)))
        return n.value

    def order(self, scene: int) -> np.ndarray:
        n = self.count(scene)
        out = np.zeros(n, np.uint64)
        m = C.c_uint32()
        self._chk(self.lib.sa_tracks_order(self.h, scene, out.ctypes.data_as(C.POINTER(C.c_uint64)), n, C.byref(m)))
        return out

    # ---- association ----
    def associate(self, scene: int, epoch: int, det: abi.sa_detections):
        ids = np.zeros(det.n, np.uint64)
        votes = np.zeros(det.n, np.uint8)
        self._chk(
            self.lib.sa_associate(self.h, scene, epoch, C.byref(det), ids.ctypes.data_as(C.POINTER(C.c_uint64)),
                                  votes.ctypes.data_as(C.POINTER(C.c_uint8)))
        )
        return ids, votes

    def host_block(self, shape, dtype=np.float32):
        """A numpy array over a pinned block from sa_host_alloc: detections' features written here go to the device without the
        staging copy.  Keep the returned array (and call host_free on it) — the block is not garbage collected."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.lib.sa_host_alloc(n)
        if not p:
            raise MemoryError("sa_host_alloc failed")
        buf = (C.c_char * n).from_address(p)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._blocks = getattr(self, "_blocks", {})
        self._blocks[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        """Frees a block from host_block().  `arr` (and every view of it) points at freed pinned memory afterwards and must not
        be touched again; the array is made read-only so that a stray write raises instead of corrupting the heap."""
        p = getattr(self, "_blocks", {}).pop(arr.ctypes.data, None)
        if p:
            try:
                arr.flags.writeable = False
            except ValueError:
                pass
            self.lib.sa_host_free(p)

    def register_device_block(self, ptr, nbytes, device=None):
        """sa_device_block_register: detections' features inside [ptr, ptr + nbytes) of device memory are read where they lie
        (pass their address as make_detections(..., feats_device_ptr=...)).  The caller keeps the memory alive, final before every
        call and untouched until the results are back."""
        self._chk(self.lib.sa_device_block_register(C.c_void_p(int(ptr)), int(nbytes), int(self.cfg.device if device is None else device)))

    def unregister_device_block(self, ptr):
        self.lib.sa_device_block_unregister(C.c_void_p(int(ptr)))

    def batch_begin(self):
        self._chk(self.lib.sa_batch_begin(self.h))

    def batch_add(self, scene: int, epoch: int, det: abi.sa_detections) -> int:
        slot = C.c_uint32()
        self._chk(self.lib.sa_batch_add(self.h, scene, epoch, C.byref(det), C.byref(slot)))
        return slot.value

    def batch_run(self):
        self._chk(self.lib.sa_batch_run(self.h))

    def batch_sync(self):
        self._chk(self.lib.sa_batch_sync(self.h))

    def batch_fetch(self, slot: int, n: int):
        ids = np.zeros(n, np.uint64)
        votes = np.zeros(n, np.uint8)
        self._chk(self.lib.sa_batch_fetch(self.h, slot, ids.ctypes.data_as(C.POINTER(C.c_uint64)), votes.ctypes.data_as(C.POINTER(C.c_uint8))))
        return ids, votes

    def batch_time(self, iters: int) -> float:
        ms = C.c_double()
        self._chk(self.lib.sa_batch_time(self.h, iters, C.byref(ms)))
        return ms.value

    # ---- whole request sets (BatchSort / BatchVisualSort::predict) ----
    @staticmethod
    def make_requests(items):
        """items: [(scene_id, epoch, sa_detections)] -> (sa_scene_request array, sa_scene_result array, [(ids, votes)] numpy outputs).
        The arrays can be handed to associate_batch / pipe_* any number of times (the outputs are overwritten)."""
        n = len(items)
        req = (abi.sa_scene_request * max(1, n))()
        res = (abi.sa_scene_result * max(1, n))()
        outs = []
        for i, (scene, epoch, det) in enumerate(items):
            req[i].scene_id, req[i].epoch, req[i].detections = scene, epoch, det
            ids, votes = np.zeros(det.n, np.uint64), np.zeros(det.n, np.uint8)
            res[i].out_track_id = ids.ctypes.data_as(C.POINTER(C.c_uint64))
            res[i].out_voting_type = votes.ctypes.data_as(C.POINTER(C.c_uint8))
            outs.append((ids, votes))
        req._keep = [d for _, _, d in items]
        res._keep = outs
        return req, res, outs

    def associate_batch(self, req, res, n=None):
        self._chk(self.lib.sa_associate_batch(self.h, len(req) if n is None else n, req, res))

    # ---- pipelined request sets: H2D of set n+1 beside the kernels of set n ----
    def pipe_stage(self, req, n=None) -> int:
        t = C.c_uint64()
        self._chk(self.lib.sa_pipe_stage(self.h, len(req) if n is None else n, req, C.byref(t)))
        return t.value

    def pipe_launch(self, ticket: int):
        self._chk(self.lib.sa_pipe_launch(self.h, ticket))

    def pipe_submit(self, req, n=None) -> int:
        t = C.c_uint64()
        self._chk(self.lib.sa_pipe_submit(self.h, len(req) if n is None else n, req, C.byref(t)))
        return t.value

    def pipe_wait(self, ticket: int, res):
        self._chk(self.lib.sa_pipe_wait(self.h, ticket, res))

    # ---- NMS (src/utils/nms.rs) ----
    def nms(self, boxes: np.ndarray, scores=None, nms_threshold: float = 0.5, score_threshold=None) -> np.ndarray:
        """Indices of the surviving boxes in the reference's output order (rank descending)."""
        boxes = np.ascontiguousarray(boxes, abi.BOX_DTYPE)
        n = len(boxes)
        keep = np.zeros(max(n, 1), np.uint32)
        m = C.c_uint32()
        sc = None if scores is None else np.ascontiguousarray(scores, np.float32)
        self._chk(self.lib.sa_nms(self.h, n, C.cast(boxes.ctypes.data, C.POINTER(abi.sa_box)),
                                  None if sc is None else sc.ctypes.data_as(C.POINTER(C.c_float)), nms_threshold,
                                  float("nan") if score_threshold is None else score_threshold,
                                  keep.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(m)))
        return keep[: m.value].copy()

    # ---- own-area shares (src/utils/clipping/bbox_own_areas.rs) ----
    def own_areas(self, boxes: np.ndarray) -> np.ndarray:
        """exclusively_owned_areas_normalized_shares of one frame's boxes (f32 per box, <= 1)."""
        boxes = np.ascontiguousarray(boxes, abi.BOX_DTYPE)
        out = np.zeros(max(len(boxes), 1), np.float32)
        self._chk(self.lib.sa_own_areas(self.h, len(boxes), C.cast(boxes.ctypes.data, C.POINTER(abi.sa_box)),
                                        out.ctypes.data_as(C.POINTER(C.c_float))))
        return out[: len(boxes)].copy()

    # ---- taps ----
    def tap_dims(self, slot: int = 0):
        n, t, k = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._chk(self.lib.sa_tap_dims(self.h, slot, C.byref(n), C.byref(t), C.byref(k)))
        return n.value, t.value, k.value

    def tap_positional(self, slot: int = 0) -> np.ndarray:
        n, t, _ = self.tap_dims(slot)
        out = np.empty((n, t), np.float32)
        self._chk(self.lib.sa_tap_positional(self.h, slot, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def tap_visual(self, slot: int = 0) -> np.ndarray:
        n, t, k = self.tap_dims(slot)
        out = np.empty((n, t, k), np.float32)
        self._chk(self.lib.sa_tap_visual(self.h, slot, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def tap_quantised(self, slot: int = 0) -> np.ndarray:
        n, t, _ = self.tap_dims(slot)
        out = np.empty((n, t), np.int64)
        self._chk(self.lib.sa_tap_quantised(self.h, slot, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def tap_track_polygons(self, scene: int) -> np.ndarray:
        """The f64 polygons of the scene's track table, [T, 4, 2] in table order."""
        t = self.count(scene)
        out = np.zeros((max(t, 1), 4, 2), np.float64)
        rows = C.c_uint32()
        self._chk(self.lib.sa_tap_track_polygons(self.h, scene, out.ctypes.data_as(C.POINTER(C.c_double)), t, C.byref(rows)))
        return out[:t]

    def tap_votes(self, slot: int = 0):
        """The BestFit vote as the frame's own first phase reduced it (engines created with SA_FLAG_TAP): (row_w, row_idx, col_w,
        col_idx, kind); kind 1 = lightest visual weight per row / column, 2 = heaviest group weight W."""
        n, t, _ = self.tap_dims(slot)
        rw, ri = np.full(n, np.nan), np.full(n, -1, np.int32)
        cw, ci = np.full(t, np.nan), np.full(t, -1, np.int32)
        kind = C.c_int32()
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        self._chk(self.lib.sa_tap_votes(self.h, slot, rw.ctypes.data_as(dp), ri.ctypes.data_as(ip), cw.ctypes.data_as(dp),
                                        ci.ctypes.data_as(ip), C.byref(kind)))
        return rw, ri, cw, ci, kind.value

    def tap_edges(self, slot: int = 0):
        """The positional vote's input as the positional tiles emitted it (SA_FLAG_TAP): (counts[N], cols[E], gains[E]) in CSR order."""
        n, _, _ = self.tap_dims(slot)
        counts = np.zeros(max(n, 1), np.uint32)
        total = C.c_uint32()
        u32p = C.POINTER(C.c_uint32)
        self._chk(self.lib.sa_tap_edges(self.h, slot, counts.ctypes.data_as(u32p), 0, None, None, C.byref(total)))
        cols = np.zeros(max(total.value, 1), np.uint32)
        gains = np.zeros(max(total.value, 1), np.int64)
        self._chk(self.lib.sa_tap_edges(self.h, slot, counts.ctypes.data_as(u32p), total.value, cols.ctypes.data_as(u32p),
                                        gains.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(total)))
        return counts[:n], cols[: total.value], gains[: total.value]

    # ---- measurement ----
    def profile_enable(self, on: bool = True):
        self._chk(self.lib.sa_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._chk(self.lib.sa_profile_reset(self.h))

    def profile_read(self) -> dict:
        arr = (abi.sa_kernel_stat * 32)()
        n = C.c_uint32()
        self._chk(self.lib.sa_profile_read(self.h, arr, 32, C.byref(n)))
        return {arr[i].name.decode(): (arr[i].launches, arr[i].total_ms) for i in range(min(n.value, 32))}

    def distance_matrix(self, kind: str, a: np.ndarray, b: np.ndarray, iters: int = 1, want_out: bool = True):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        n, d = a.shape
        t, d2 = b.shape
        assert d == d2
        out = np.empty((n, t), np.float32) if want_out else None
        ms = C.c_double()
        fp = C.POINTER(C.c_float)
        self._chk(
            self.lib.sa_feature_distance_matrix(
                self.h, abi.SA_VIS_COSINE if kind == "cosine" else abi.SA_VIS_EUCLIDEAN, n, t, d, a.ctypes.data_as(fp),
                b.ctypes.data_as(fp), out.ctypes.data_as(fp) if want_out else None, iters, C.byref(ms))
        )
        return out, ms.value


class Cluster:
    """One process, one engine per GPU (include/similari_assoc.h: sa_cluster_*): scenes routed by scene_id % n_shards, every shard's
    share of a request set running on its own device concurrently with the others."""

    def __init__(self, cfg: abi.sa_config, devices=None, n_shards=None, lib: C.CDLL | None = None):
        self.lib = lib or abi.load_library()
        if devices is None:
            devices = list(range(n_shards or 1))
        self.devices = list(devices)
        arr = (C.c_int32 * len(self.devices))(*self.devices)
        self.h = C.c_void_p()
        rc = self.lib.sa_cluster_create(C.byref(cfg), len(self.devices), arr, C.byref(self.h))
        if rc != abi.SA_OK:
            msg = self.lib.sa_cluster_last_error(None)
            raise EngineError(rc, msg.decode() if msg else "")
        self.cfg = cfg

    def close(self):
        if self.h:
            self.lib.sa_cluster_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != abi.SA_OK:
            msg = self.lib.sa_cluster_last_error(self.h)
            raise EngineError(rc, msg.decode() if msg else "")

    def __len__(self):
        return self.lib.sa_cluster_size(self.h)

    def shard_of(self, scene: int) -> int:
        return self.lib.sa_cluster_shard_of(self.h, scene)

    def engine(self, shard: int) -> Engine:
        return Engine.borrowed(self.lib, self.lib.sa_cluster_engine(self.h, shard), self.cfg)

    def upsert(self, scene: int, tracks: abi.sa_tracks):
        self._chk(self.lib.sa_cluster_tracks_upsert(self.h, scene, C.byref(tracks)))

    def remove(self, scene: int, ids):
        ids = np.ascontiguousarray(ids, np.uint64)
        self._chk(self.lib.sa_cluster_tracks_remove(self.h, scene, len(ids), ids.ctypes.data_as(C.POINTER(C.c_uint64))))

    def associate_batch(self, req, res, n=None):
        self._chk(self.lib.sa_cluster_associate_batch(self.h, len(req) if n is None else n, req, res))

    def last_ms(self):
        return [self.lib.sa_cluster_last_ms(self.h, k) for k in range(len(self))]
