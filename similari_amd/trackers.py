"""Python mirror of the reference's tracker classes (the names its PyO3 module exports), bound to the C ABI of
include/similari_tracker.h.  Association runs on the GPU; nothing here computes the hot path.

    Sort / BatchSort                      src/trackers/sort/simple_api.rs, sort/batch_api.rs
    VisualSort / BatchVisualSort          src/trackers/visual_sort/simple_api.rs, visual_sort/batch_api.rs
    VisualSortOptions, VisualSortObservation, PositionalMetricType, VisualSortMetricType,
    SpatioTemporalConstraints, Universal2DBox, BoundingBox, SortTrack, VotingType, PredictionBatchRequest
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import abi


class VotingType:
    Visual = abi.SA_VOTE_VISUAL
    Positional = abi.SA_VOTE_POSITIONAL


class PositionalMetricType:
    def __init__(self, kind, threshold=0.3):
        self.kind, self.threshold = kind, threshold

    @staticmethod
    def iou(threshold: float = 0.3):
        return PositionalMetricType(abi.SA_POS_IOU, threshold)

    @staticmethod
    def maha():
        return PositionalMetricType(abi.SA_POS_MAHALANOBIS, 0.0)


class VisualSortMetricType:
    def __init__(self, kind, threshold):
        self.kind, self.threshold = kind, threshold

    @staticmethod
    def euclidean(threshold: float):
        assert threshold > 0.0, "Threshold must be a positive number"
        return VisualSortMetricType(abi.SA_VIS_EUCLIDEAN, threshold)

    @staticmethod
    def cosine(threshold: float):
        assert -1.0 <= threshold <= 1.0, "Threshold must lay within [-1.0:1:0]"
        return VisualSortMetricType(abi.SA_VIS_COSINE, threshold)


class SpatioTemporalConstraints:
    def __init__(self):
        self.constraints = []

    def add_constraints(self, constraints):
        for delta, max_distance in constraints:
            assert max_distance > 0.0, "The distance is expected to be a positive float"
            self.constraints.append((int(delta), float(max_distance)))
        self.constraints.sort(key=lambda c: c[0])  # stable
        dedup = []
        for c in self.constraints:
            if not dedup or dedup[-1][0] != c[0]:
                dedup.append(c)
        self.constraints = dedup
        return self


@dataclass
class Universal2DBox:
    xc: float
    yc: float
    angle: Optional[float]
    aspect: float
    height: float
    confidence: float = 1.0

    @staticmethod
    def new_with_confidence(xc, yc, angle, aspect, height, confidence):
        assert 0.0 <= confidence <= 1.0, "Confidence must lay between 0.0 and 1.0"
        return Universal2DBox(xc, yc, angle, aspect, height, confidence)

    def to_c(self) -> abi.sa_box:
        b = abi.sa_box()
        b.xc, b.yc, b.aspect, b.height, b.confidence = self.xc, self.yc, self.aspect, self.height, self.confidence
        b.has_angle = 0 if self.angle is None else 1
        b.angle = 0.0 if self.angle is None else self.angle
        return b

    @staticmethod
    def from_c(b: abi.sa_box) -> "Universal2DBox":
        return Universal2DBox(b.xc, b.yc, b.angle if b.has_angle else None, b.aspect, b.height, b.confidence)


@dataclass
class BoundingBox:
    left: float
    top: float
    width: float
    height: float
    confidence: float = 1.0

    def as_xyaah(self) -> Universal2DBox:  # bbox.rs:246-257, f32 arithmetic
        f = np.float32
        l, t, w, h = f(self.left), f(self.top), f(self.width), f(self.height)
        return Universal2DBox(float(l + w / f(2.0)), float(t + h / f(2.0)), None, float(w / h), float(h), self.confidence)


@dataclass
class SortTrack:
    id: int
    epoch: int
    predicted_bbox: Universal2DBox
    observed_bbox: Universal2DBox
    scene_id: int
    length: int
    voting_type: int
    custom_object_id: Optional[int]

    @staticmethod
    def from_c(t: abi.sa_sort_track) -> "SortTrack":
        return SortTrack(t.id, t.epoch, Universal2DBox.from_c(t.predicted_bbox), Universal2DBox.from_c(t.observed_bbox),
                         t.scene_id, t.length, t.voting_type, t.custom_object_id if t.has_custom_object_id else None)


class VisualSortObservation:
    def __init__(self, feature, feature_quality, bounding_box, custom_object_id=None, own_area=None):
        self.feature = None if feature is None else np.ascontiguousarray(feature, np.float32)
        self.feature_quality = feature_quality
        self.bounding_box = bounding_box
        self.custom_object_id = custom_object_id
        self.own_area = own_area  # the caller supplies exclusively_owned_areas shares (out of scope here)


class VisualSortOptions:
    """Builder with the reference's defaults (visual_sort/options.rs:194-205, metric/builder.rs:26-42)."""

    def __init__(self):
        self._max_idle_epochs = 2
        self._kept_history_length = 10
        self._visual_metric = VisualSortMetricType(abi.SA_VIS_EUCLIDEAN, 3.4028234663852886e38)
        self._positional_metric = PositionalMetricType.iou(0.3)
        self._visual_minimal_track_length = 3
        self._visual_minimal_area = 0.0
        self._visual_minimal_quality_use = 0.0
        self._visual_minimal_quality_collect = 0.0
        self._visual_max_observations = 5
        self._visual_min_votes = 1
        self._own_use = 0.0
        self._own_collect = 0.0
        self._positional_min_confidence = 0.1
        self._constraints = SpatioTemporalConstraints()
        self._pw, self._vw = 1.0 / 20.0, 1.0 / 160.0

    def max_idle_epochs(self, n): self._max_idle_epochs = n; return self
    def kept_history_length(self, n): self._kept_history_length = n; return self
    def visual_metric(self, m): self._visual_metric = m; return self
    def positional_metric(self, m): self._positional_metric = m; return self
    def visual_minimal_track_length(self, n): assert n > 0; self._visual_minimal_track_length = n; return self
    def visual_minimal_area(self, a): assert a >= 0.0; self._visual_minimal_area = a; return self
    def visual_minimal_quality_use(self, q): assert q >= 0.0; self._visual_minimal_quality_use = q; return self
    def visual_minimal_quality_collect(self, q): assert q >= 0.0; self._visual_minimal_quality_collect = q; return self
    def visual_max_observations(self, n): self._visual_max_observations = n; return self
    def visual_min_votes(self, n): self._visual_min_votes = n; return self
    def visual_minimal_own_area_percentage_use(self, a): assert 0.0 <= a <= 1.0; self._own_use = a; return self
    def visual_minimal_own_area_percentage_collect(self, a): assert 0.0 <= a <= 1.0; self._own_collect = a; return self
    def positional_min_confidence(self, c): self._positional_min_confidence = c; return self
    def spatio_temporal_constraints(self, c): self._constraints = c; return self
    def kalman_position_weight(self, w): self._pw = w; return self
    def kalman_velocity_weight(self, w): self._vw = w; return self


class TrackerError(RuntimeError):
    pass


class _Tracker:
    def __init__(self, opts: abi.sa_tracker_options, keep, lib=None):
        self.lib = lib or abi.load_library()
        self._keep = keep
        self.opts = opts
        self.h = C.c_void_p()
        rc = self.lib.sa_tracker_create(C.byref(opts), C.byref(self.h))
        if rc != abi.SA_OK:
            raise TrackerError(f"sa_tracker_create failed ({rc}): {self.lib.sa_tracker_last_error(None).decode()}")

    def close(self):
        if self.h:
            self.lib.sa_tracker_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != abi.SA_OK:
            raise TrackerError(f"similari_tracker error {rc}: {self.lib.sa_tracker_last_error(self.h).decode()}")

    def _obs_array(self, items, keep):
        arr = (abi.sa_observation * max(1, len(items)))()
        D = self.opts.feature_len
        for i, it in enumerate(items):
            if isinstance(it, VisualSortObservation):
                box, feat, q, cid, own = it.bounding_box, it.feature, it.feature_quality, it.custom_object_id, it.own_area
            else:
                box, cid = it
                feat, q, own = None, None, None
            o = arr[i]
            if isinstance(box, BoundingBox):  # bbox.rs:246-257: ltwh -> xyaah in f32; bbox.rs:123-126: confidence in [0, 1]
                if not (0.0 <= box.confidence <= 1.0):
                    raise TrackerError(f"confidence {box.confidence} must lie within [0.0, 1.0] (bbox.rs:123-126)")
                box = box.as_xyaah()
            o.bbox = box.to_c() if isinstance(box, Universal2DBox) else box
            if feat is not None and self.opts.visual:
                assert len(feat) == D, f"feature length {len(feat)} != {D}"
                keep.append(feat)
                o.feature = feat.ctypes.data_as(C.POINTER(C.c_float))
            o.feature_quality = math.nan if q is None else q
            o.own_area = math.nan if own is None else own
            o.has_custom_object_id = 0 if cid is None else 1
            o.custom_object_id = 0 if cid is None else cid
        return arr

    def predict_with_scene(self, scene_id: int, items: Sequence):
        keep = []
        arr = self._obs_array(items, keep)
        out = (abi.sa_sort_track * max(1, len(items)))()
        self._chk(self.lib.sa_tracker_predict(self.h, scene_id, len(items), arr, out))
        return [SortTrack.from_c(out[i]) for i in range(len(items))]

    def predict(self, items: Sequence):
        return self.predict_with_scene(0, items)

    def predict_batch(self, batch: "PredictionBatchRequest"):
        scenes = list(batch.scenes.keys())
        keep = []
        arrs = [self._obs_array(batch.scenes[s], keep) for s in scenes]
        outs = [(abi.sa_sort_track * max(1, len(batch.scenes[s])))() for s in scenes]
        ids = (C.c_uint64 * max(1, len(scenes)))(*scenes)
        counts = (C.c_uint32 * max(1, len(scenes)))(*[len(batch.scenes[s]) for s in scenes])
        pa = (C.POINTER(abi.sa_observation) * max(1, len(scenes)))(*[C.cast(a, C.POINTER(abi.sa_observation)) for a in arrs])
        po = (C.POINTER(abi.sa_sort_track) * max(1, len(scenes)))(*[C.cast(o, C.POINTER(abi.sa_sort_track)) for o in outs])
        self._chk(self.lib.sa_tracker_predict_batch(self.h, len(scenes), ids, counts, pa, po))
        return {s: [SortTrack.from_c(outs[k][i]) for i in range(len(batch.scenes[s]))] for k, s in enumerate(scenes)}

    def predict_batch_async(self, batch: "PredictionBatchRequest") -> "PredictionBatchResult":
        """Batch*::predict as the reference shapes it: returns once the request set is on the device; the handle delivers scene by scene."""
        scenes = list(batch.scenes.keys())
        keep = []
        arrs = [self._obs_array(batch.scenes[s], keep) for s in scenes]
        ids = (C.c_uint64 * max(1, len(scenes)))(*scenes)
        counts = (C.c_uint32 * max(1, len(scenes)))(*[len(batch.scenes[s]) for s in scenes])
        pa = (C.POINTER(abi.sa_observation) * max(1, len(scenes)))(*[C.cast(a, C.POINTER(abi.sa_observation)) for a in arrs])
        h = C.c_void_p()
        self._chk(self.lib.sa_tracker_predict_batch_begin(self.h, len(scenes), ids, counts, pa, C.byref(h)))
        return PredictionBatchResult(self, h, max([len(batch.scenes[s]) for s in scenes] + [1]), keep)

    def idle_tracks_with_scene(self, scene_id: int):
        n = C.c_uint32()
        self._chk(self.lib.sa_tracker_idle_tracks(self.h, scene_id, None, 0, C.byref(n)))
        out = (abi.sa_sort_track * max(1, n.value))()
        self._chk(self.lib.sa_tracker_idle_tracks(self.h, scene_id, out, n.value, C.byref(n)))
        return [SortTrack.from_c(out[i]) for i in range(n.value)]

    def idle_tracks(self):
        return self.idle_tracks_with_scene(0)

    def skip_epochs_for_scene(self, scene_id: int, n: int):
        self._chk(self.lib.sa_tracker_skip_epochs(self.h, scene_id, n))

    def skip_epochs(self, n: int):
        self.skip_epochs_for_scene(0, n)

    def current_epoch_with_scene(self, scene_id: int) -> int:
        e = C.c_uint64()
        self._chk(self.lib.sa_tracker_current_epoch(self.h, scene_id, C.byref(e)))
        return e.value

    def current_epoch(self) -> int:
        return self.current_epoch_with_scene(0)

    def wasted(self):
        n = C.c_uint32()
        self._chk(self.lib.sa_tracker_wasted(self.h, None, 0, C.byref(n)))
        out = (abi.sa_sort_track * max(1, n.value))()
        self._chk(self.lib.sa_tracker_wasted(self.h, out, n.value, C.byref(n)))
        return [SortTrack.from_c(out[i]) for i in range(n.value)]

    def wasted_count(self) -> int:
        n = C.c_uint32()
        self._chk(self.lib.sa_tracker_wasted(self.h, None, 0, C.byref(n)))
        return n.value

    def clear_wasted(self):
        self._chk(self.lib.sa_tracker_clear_wasted(self.h))

    def active_tracks(self) -> int:
        n = C.c_uint64()
        self._chk(self.lib.sa_tracker_active_tracks(self.h, C.byref(n)))
        return n.value

    def track_info(self, track_id: int) -> dict:
        out = (C.c_uint64 * 4)()
        self._chk(self.lib.sa_tracker_track_info(self.h, track_id, out))
        return dict(visual_features_collected_count=out[0], observations=out[1], history=out[2], track_length=out[3])

    def track_state(self, track_id: int):
        m = np.zeros(10, np.float32)
        c = np.zeros(100, np.float32)
        fp = C.POINTER(C.c_float)
        self._chk(self.lib.sa_tracker_track_state(self.h, track_id, m.ctypes.data_as(fp), c.ctypes.data_as(fp)))
        return m, c


def sort_options(bbox_history, max_idle_epochs, method, min_confidence, constraints, pw, vw, batch=False, device=-1,
                 device_upkeep=False, workers=0, devices=None, spin_us=-1):
    keep = abi.Keep()
    o = abi.sa_tracker_options()
    o.struct_size = C.sizeof(abi.sa_tracker_options)
    o.device = device
    o.visual = 0
    o.batch_ids = 1 if batch else 0
    o.history_length = bbox_history
    o.auto_waste_periodicity = 100
    o.max_idle_epochs = max_idle_epochs
    o.positional_kind = method.kind
    o.positional_threshold = method.threshold
    o.positional_min_confidence = min_confidence
    cons = constraints.constraints if constraints is not None else []
    o.n_constraints = len(cons)
    d = keep.arr([c[0] for c in cons], np.uint64) if cons else None
    m = keep.arr([c[1] for c in cons], np.float32) if cons else None
    o.constraint_epoch_delta = abi._ptr(d, C.c_uint64)
    o.constraint_max_dist = abi._ptr(m, C.c_float)
    o.kalman_position_weight = pw
    o.kalman_velocity_weight = vw
    o.device_upkeep = 1 if device_upkeep else 0
    o.workers = workers
    # devices = [ordinals]: one engine per entry, scenes dealt out scene_id % len(devices) (include/similari_tracker.h)
    dv = keep.arr(list(devices), np.int32) if devices else None
    o.n_devices = len(devices) if devices else 0
    o.devices = abi._ptr(dv, C.c_int32)
    o.spin_us = spin_us
    return o, keep


def visual_options(opts: VisualSortOptions, feature_len: int, batch=False, device=-1, device_upkeep=False, workers=0, devices=None, spin_us=-1):
    o, keep = sort_options(opts._kept_history_length, opts._max_idle_epochs, opts._positional_metric,
                           opts._positional_min_confidence, opts._constraints, opts._pw, opts._vw, batch, device, device_upkeep, workers,
                           devices, spin_us)
    o.visual = 1
    o.visual_kind = opts._visual_metric.kind
    o.visual_threshold = opts._visual_metric.threshold
    o.feature_len = feature_len
    o.visual_max_observations = opts._visual_max_observations
    o.visual_min_votes = opts._visual_min_votes
    o.visual_minimal_track_length = opts._visual_minimal_track_length
    o.visual_minimal_area = opts._visual_minimal_area
    o.visual_minimal_quality_use = opts._visual_minimal_quality_use
    o.visual_minimal_quality_collect = opts._visual_minimal_quality_collect
    o.visual_minimal_own_area_percentage_use = opts._own_use
    o.visual_minimal_own_area_percentage_collect = opts._own_collect
    return o, keep


class Sort(_Tracker):
    """Sort::new(shards, bbox_history, max_idle_epochs, method, min_confidence, spatio_temporal_constraints, pos_w, vel_w).
    `shards` is accepted for signature compatibility: the store is one GPU-resident table per scene."""

    def __init__(self, shards=1, bbox_history=1, max_idle_epochs=5, method=None, min_confidence=0.05,
                 spatio_temporal_constraints=None, kalman_position_weight=1.0 / 20.0, kalman_velocity_weight=1.0 / 160.0,
                 device=-1, _batch=False, device_upkeep=False, workers=0, devices=None, spin_us=-1):
        assert bbox_history > 0
        o, keep = sort_options(bbox_history, max_idle_epochs, method or PositionalMetricType.iou(0.3), min_confidence,
                               spatio_temporal_constraints, kalman_position_weight, kalman_velocity_weight, _batch, device,
                               device_upkeep, workers, devices, spin_us)
        super().__init__(o, keep)


class VisualSort(_Tracker):
    """VisualSort::new(shards, &VisualSortOptions); `feature_len` fixes the engine's feature dimension."""

    def __init__(self, shards=1, opts: Optional[VisualSortOptions] = None, feature_len: int = 0, device=-1, _batch=False,
                 device_upkeep=False, workers=0, devices=None, spin_us=-1):
        o, keep = visual_options(opts or VisualSortOptions(), feature_len, _batch, device, device_upkeep, workers, devices, spin_us)
        super().__init__(o, keep)


class PredictionBatchResult:
    """trackers/batch.rs:19-38: ready() / get() -> (scene_id, [SortTrack]) / batch_size()."""

    def __init__(self, tracker, handle, cap, keep):
        self._t, self._h, self._cap, self._keep = tracker, handle, cap, keep

    def batch_size(self) -> int:
        return self._t.lib.sa_batch_result_size(self._h)

    def ready(self) -> bool:
        return bool(self._t.lib.sa_batch_result_ready(self._h))

    def get(self):
        """The next finished scene: (scene id, its tracks).  Through sa_batch_result_take — the tracks are read where the handle holds
        them (the reference's get() moves the scene's Vec out of the channel; sa_batch_result_get is the copying form for C callers)."""
        ptr = C.POINTER(abi.sa_sort_track)()
        sid, n = C.c_uint64(), C.c_uint32()
        rc = self._t.lib.sa_batch_result_take(self._h, C.byref(sid), C.byref(ptr), C.byref(n))
        if rc != abi.SA_OK:
            raise TrackerError(f"sa_batch_result_take failed ({rc}): {self._t.lib.sa_tracker_last_error(None).decode()}")
        return sid.value, [SortTrack.from_c(ptr[i]) for i in range(n.value)]

    def get_copy(self):
        """The same through sa_batch_result_get (the scene's tracks copied into an array of the caller's)."""
        out = (abi.sa_sort_track * max(1, self._cap))()
        sid, n = C.c_uint64(), C.c_uint32()
        rc = self._t.lib.sa_batch_result_get(self._h, C.byref(sid), out, self._cap, C.byref(n))
        if rc != abi.SA_OK:
            raise TrackerError(f"sa_batch_result_get failed ({rc}): {self._t.lib.sa_tracker_last_error(None).decode()}")
        return sid.value, [SortTrack.from_c(out[i]) for i in range(n.value)]

    def close(self):
        if self._h:
            self._t.lib.sa_batch_result_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PredictionBatchRequest:
    """trackers/batch.rs: scene -> observations; results come back per scene."""

    def __init__(self):
        self.scenes = {}

    def add(self, scene_id: int, elt):
        self.scenes.setdefault(scene_id, []).append(elt)


class BatchSort(Sort):
    """BatchSort::new(distance_shards, voting_shards, ...): `voting_shards` = the threads that work on the scenes of a request set
    (0 = the facade's choice); `distance_shards` is accepted for signature compatibility (the distances are the GPU's)."""

    def __init__(self, distance_shards=1, voting_shards=0, **kw):
        super().__init__(shards=distance_shards, _batch=True, workers=voting_shards, **kw)

    def predict(self, batch: PredictionBatchRequest):
        return self.predict_batch(batch)


class BatchVisualSort(VisualSort):
    def __init__(self, distance_shards=1, voting_shards=0, opts=None, feature_len=0, device=-1, device_upkeep=False, devices=None, spin_us=-1):
        super().__init__(shards=distance_shards, opts=opts, feature_len=feature_len, device=device, _batch=True,
                         device_upkeep=device_upkeep, workers=voting_shards, devices=devices, spin_us=spin_us)

    def predict(self, batch: PredictionBatchRequest):
        return self.predict_batch(batch)
