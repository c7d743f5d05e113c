"""Scene sharding of the batched trackers (BatchSort / BatchVisualSort) over the GPUs of one node.

Scenes are independent units on this path: `compatible()` is false across scene ids (src/trackers/sort.rs:251,
visual_sort/track_attributes.rs:189), and the reference already fans the scenes of one PredictionBatchRequest out to voting
threads (sort/batch_api.rs:278-288).  Here a scene is owned by rank `scene_id % world` — sticky, so the scene's track table
(boxes, Kalman projection, feature bank) stays resident in that GPU's HBM, the way `track_id % shards` pins a track to a
store shard in the reference (track/store.rs:490-493).

There is NO data-path collective: no cost cell or assignment depends on another scene.  The only exchange is the request
scatter and the result gather of one `predict(batch)` call — KB-scale, latency-bound — done with `torch.distributed`
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests):

    root                         every rank
    ----                         ----------
    partition scenes by owner
    broadcast  header[world][2]  (request bytes, observation count per rank)
    scatter    request bytes  ->  decode, local tracker.predict_batch (one set of launches for all local scenes)
    gather     result records <-  encode SortTrack records
    reassemble {scene: [SortTrack]}

Track ids: each rank's tracker counts 1, 2, 3, ... on its own; the global id is `(local - 1) * world + rank + 1`, unique
across ranks and equal to the local id when world == 1 (the reference draws ids from one shared counter whose interleaving
across scenes is timing-dependent, sort/batch_api.rs:102-106, so only uniqueness is contractual).

The local tracker is anything with `predict_batch(PredictionBatchRequest) -> {scene: [SortTrack]}`: `BatchSort` /
`BatchVisualSort` of similari_amd.trackers in production.  This module computes nothing on the hot path."""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np

from . import trackers as TR

OBS_WIRE = np.dtype([("xc", "<f4"), ("yc", "<f4"), ("angle", "<f4"), ("aspect", "<f4"), ("height", "<f4"), ("confidence", "<f4"),
                     ("has_angle", "<i4"), ("has_feature", "<i4"), ("quality", "<f4"), ("own_area", "<f4"),
                     ("has_cid", "<i4"), ("pad", "<i4"), ("cid", "<i8")])          # 56 B
TRACK_WIRE = np.dtype([("id", "<u8"), ("epoch", "<u8"), ("scene_id", "<u8"), ("length", "<u8"), ("cid", "<i8"),
                       ("voting_type", "<i4"), ("has_cid", "<i4"),
                       ("pred", "<f4", (6,)), ("obs", "<f4", (6,)), ("pred_has_angle", "<i4"), ("obs_has_angle", "<i4")])  # 104 B
SHUTDOWN = -1


def owner(scene_id: int, world: int) -> int:
    return int(scene_id) % world


def global_id(local_id: int, rank: int, world: int) -> int:
    return (int(local_id) - 1) * world + rank + 1 if local_id else 0


def partition(batch: "TR.PredictionBatchRequest", world: int) -> List[Dict[int, list]]:
    """Scenes of one request -> per-rank sub-requests, scene order preserved."""
    parts: List[Dict[int, list]] = [dict() for _ in range(world)]
    for scene, items in batch.scenes.items():
        parts[owner(scene, world)][scene] = items
    return parts


# ---- wire format ---------------------------------------------------------------------------------------------------------
def _box_fields(rec, box):
    if isinstance(box, TR.BoundingBox):
        box = box.as_xyaah()
    rec["xc"], rec["yc"], rec["aspect"], rec["height"], rec["confidence"] = box.xc, box.yc, box.aspect, box.height, box.confidence
    rec["has_angle"] = 0 if box.angle is None else 1
    rec["angle"] = 0.0 if box.angle is None else box.angle


def encode_request(scenes: Dict[int, list], feature_len: int) -> np.ndarray:
    """{scene: [VisualSortObservation | (box, custom_id)]} -> one uint8 buffer:
    int64 header [n_scenes, n_obs, n_feat_rows, D] | int64 [n_scenes][2] (scene, count) | OBS_WIRE[n_obs] | f32 [n_feat_rows][D]"""
    ids = list(scenes.keys())
    n_obs = sum(len(scenes[s]) for s in ids)
    obs = np.zeros(n_obs, OBS_WIRE)
    feats = []
    i = 0
    for s in ids:
        for it in scenes[s]:
            r = obs[i]
            if isinstance(it, TR.VisualSortObservation):
                _box_fields(r, it.bounding_box)
                r["quality"] = math.nan if it.feature_quality is None else it.feature_quality
                r["own_area"] = math.nan if it.own_area is None else it.own_area
                cid = it.custom_object_id
                if it.feature is not None:
                    assert len(it.feature) == feature_len, f"feature length {len(it.feature)} != {feature_len}"
                    r["has_feature"] = 1
                    feats.append(np.asarray(it.feature, np.float32))
            else:
                box, cid = it
                _box_fields(r, box)
                r["quality"] = math.nan
                r["own_area"] = math.nan
            r["has_cid"] = 0 if cid is None else 1
            r["cid"] = 0 if cid is None else cid
            i += 1
    head = np.array([len(ids), n_obs, len(feats), feature_len], np.int64)
    table = np.array([[s, len(scenes[s])] for s in ids], np.int64).reshape(len(ids), 2)
    fm = np.stack(feats).astype(np.float32) if feats else np.zeros((0, max(feature_len, 1)), np.float32)
    return np.concatenate([head.view(np.uint8), table.reshape(-1).view(np.uint8), obs.view(np.uint8), fm.reshape(-1).view(np.uint8)])


def decode_request(buf: np.ndarray) -> "TR.PredictionBatchRequest":
    buf = np.ascontiguousarray(buf, np.uint8)
    n_scenes, n_obs, n_feat, D = (int(x) for x in buf[:32].view(np.int64))
    o = 32
    table = buf[o:o + 16 * n_scenes].view(np.int64).reshape(n_scenes, 2)
    o += 16 * n_scenes
    obs = buf[o:o + OBS_WIRE.itemsize * n_obs].view(OBS_WIRE)
    o += OBS_WIRE.itemsize * n_obs
    feats = buf[o:o + 4 * n_feat * D].view(np.float32).reshape(n_feat, D) if n_feat else None
    req = TR.PredictionBatchRequest()
    i = f = 0
    for scene, count in table:
        req.scenes[int(scene)] = []
        for _ in range(int(count)):
            r = obs[i]
            box = TR.Universal2DBox(float(r["xc"]), float(r["yc"]), float(r["angle"]) if r["has_angle"] else None, float(r["aspect"]),
                                    float(r["height"]), float(r["confidence"]))
            cid = int(r["cid"]) if r["has_cid"] else None
            if D:
                ft = None
                if r["has_feature"]:
                    ft = feats[f].copy()
                    f += 1
                q = None if math.isnan(float(r["quality"])) else float(r["quality"])
                oa = None if math.isnan(float(r["own_area"])) else float(r["own_area"])
                req.scenes[int(scene)].append(TR.VisualSortObservation(ft, q, box, cid, oa))
            else:
                req.scenes[int(scene)].append((box, cid))
            i += 1
    return req


def _b6(b: "TR.Universal2DBox"):
    return [b.xc, b.yc, 0.0 if b.angle is None else b.angle, b.aspect, b.height, b.confidence], (0 if b.angle is None else 1)


def encode_results(results: Dict[int, list], order: List[int], rank: int, world: int) -> np.ndarray:
    n = sum(len(results[s]) for s in order)
    out = np.zeros(n, TRACK_WIRE)
    i = 0
    for s in order:
        for t in results[s]:
            r = out[i]
            r["id"] = global_id(t.id, rank, world)
            r["epoch"], r["scene_id"], r["length"], r["voting_type"] = t.epoch, t.scene_id, t.length, t.voting_type
            r["has_cid"] = 0 if t.custom_object_id is None else 1
            r["cid"] = 0 if t.custom_object_id is None else t.custom_object_id
            r["pred"], r["pred_has_angle"] = _b6(t.predicted_bbox)
            r["obs"], r["obs_has_angle"] = _b6(t.observed_bbox)
            i += 1
    return out.view(np.uint8)


def decode_results(buf: np.ndarray) -> List["TR.SortTrack"]:
    recs = np.ascontiguousarray(buf, np.uint8).view(TRACK_WIRE)
    out = []
    for r in recs:
        def bx(v, has):
            return TR.Universal2DBox(float(v[0]), float(v[1]), float(v[2]) if has else None, float(v[3]), float(v[4]), float(v[5]))
        out.append(TR.SortTrack(int(r["id"]), int(r["epoch"]), bx(r["pred"], r["pred_has_angle"]), bx(r["obs"], r["obs_has_angle"]),
                                int(r["scene_id"]), int(r["length"]), int(r["voting_type"]), int(r["cid"]) if r["has_cid"] else None))
    return out


# ---- the collective call ---------------------------------------------------------------------------------------------------
class ShardedBatchTracker:
    """SPMD wrapper: every rank constructs it around its local tracker and calls `predict` in lockstep; the root passes
    the PredictionBatchRequest and gets {scene: [SortTrack]} back, the other ranks pass None and get None.
    Workers that have nothing else to do call `serve_forever()`; the root ends them with `shutdown()`."""

    def __init__(self, local_tracker, feature_len: int = 0, group=None, root: int = 0, device=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.local = local_tracker
        self.feature_len = int(feature_len)
        self.group = group
        self.root = root
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        self.device = device

    def owner(self, scene_id: int) -> int:
        return owner(scene_id, self.world)

    def _t(self, arr: np.ndarray, size: int):
        t = self.torch.zeros(size, dtype=self.torch.uint8)
        if len(arr):
            t[: len(arr)] = self.torch.from_numpy(np.ascontiguousarray(arr))
        return t.to(self.device)

    def predict(self, batch: Optional["TR.PredictionBatchRequest"] = None):
        """Collective.  Root: pass the request, get {scene: [SortTrack]}.  Other ranks: pass None, get None."""
        return self._step(batch, False)[1]

    def _step(self, batch, shutdown: bool):
        torch, dist = self.torch, self.dist
        is_root = self.rank == self.root
        header = torch.zeros((self.world, 2), dtype=torch.int64)
        bufs, orders = None, None
        if is_root:
            if shutdown:
                header[:, 0] = SHUTDOWN
            else:
                parts = partition(batch, self.world)
                orders = [list(p.keys()) for p in parts]
                bufs = [encode_request(p, self.feature_len) for p in parts]
                for r in range(self.world):
                    header[r, 0] = len(bufs[r])
                    header[r, 1] = sum(len(v) for v in parts[r].values())
        header = header.to(self.device)
        dist.broadcast(header, src=self.root, group=self.group)
        header = header.cpu()
        if int(header[0, 0]) == SHUTDOWN:
            return "shutdown", None
        req_max = int(header[:, 0].max())
        res_max = int(header[:, 1].max()) * TRACK_WIRE.itemsize
        # scatter the encoded sub-requests (padded to the longest: collectives want equal shapes)
        mine = torch.zeros(req_max, dtype=torch.uint8, device=self.device)
        dist.scatter(mine, [self._t(b, req_max) for b in bufs] if is_root else None, src=self.root, group=self.group)
        sub = decode_request(mine.cpu().numpy()[: int(header[self.rank, 0])])
        local = self.local.predict_batch(sub) if sub.scenes else {}
        enc = encode_results(local, list(sub.scenes.keys()), self.rank, self.world)
        out = self._t(enc, max(res_max, 1))
        gathered = [torch.zeros(max(res_max, 1), dtype=torch.uint8, device=self.device) for _ in range(self.world)] if is_root else None
        dist.gather(out, gathered, dst=self.root, group=self.group)
        if not is_root:
            return "served", None
        result: Dict[int, list] = {}
        for r in range(self.world):
            n = int(header[r, 1])
            tracks = decode_results(gathered[r].cpu().numpy()[: n * TRACK_WIRE.itemsize])
            i = 0
            for s in orders[r]:
                c = len(batch.scenes[s])
                result[s] = tracks[i:i + c]
                i += c
        return "served", {s: result[s] for s in batch.scenes}  # request order

    def serve_forever(self):
        """Worker loop: take part in the root's predict() calls until it calls shutdown()."""
        assert self.rank != self.root
        while self._step(None, False)[0] != "shutdown":
            pass

    def shutdown(self):
        assert self.rank == self.root
        self._step(None, True)


# ---- array-level sharding of the association itself (no per-observation Python) ------------------------------------------
# The tracker-level wrapper above moves Python objects; this one moves the arrays the C ABI takes.  One request set =
# [(scene_id, epoch, boxes[N] (abi.BOX_DTYPE), feats[N, D] | None, quality[N] | None)]; the root packs every rank's share into one
# byte buffer with numpy concatenations, ONE scatter delivers them, every rank hands its share to its engine as one
# sa_associate_batch (one set of launches), ONE gather returns ids[N] + votes[N] per scene.  Buffers have a fixed capacity agreed at
# construction (collectives want equal shapes), so there is no size exchange per call.
#
# A share = a small PREFIX (head, scene table, boxes, qualities: kilobytes) and the BULK (the detections' feature rows), which starts
# at a fixed offset.  On GPUs (RCCL) the share lands in device memory and the bulk STAYS there: the rank registers its receive buffer
# with the engine once (sa_device_block_register) and hands the feature rows over as device pointers into it — only the prefix is
# copied to the host (one small D2H), instead of the whole share going GPU -> host -> GPU.
ASSOC_HEAD = 4  # int64 words: n_scenes, D, total N, flags (1 = shutdown, 4 = the root refused the request set: every rank raises)
FLAG_SHUTDOWN, FLAG_ABORT = 1, 4


class RequestRefused(RuntimeError):
    """The root refused a request set BEFORE the first collective (a share exceeds the capacities agreed at construction): every rank
    learns it from its share's flags and skips the set consistently — the only exception a worker loop may swallow."""


class ShardFailed(RuntimeError):
    """A rank's engine call failed between the scatter and the gather: the rank still takes part in the gather (with an error marker
    in its result block), the root raises this, and the failing rank re-raises its own exception afterwards."""


def prefix_bytes(max_scenes: int, capacity_rows: int) -> int:
    """Where the feature rows of a share start: after head, table, boxes and qualities at full capacity, on a 256-byte boundary."""
    from . import abi

    return (32 + 32 * int(max_scenes) + int(capacity_rows) * (abi.BOX_DTYPE.itemsize + 4) + 255) // 256 * 256


def pack_share(items, D: int, feat_base: int = 0, flags: int = 0) -> np.ndarray:
    """[(scene, epoch, boxes, feats, quality)] -> uint8 buffer: int64 head[4] | int64 table[n][4] (scene, epoch, N, has_feats |
    has_quality << 1) | boxes | quality | (zero padding up to feat_base) | feats."""
    from . import abi

    n = len(items)
    table = np.zeros((n, 4), np.int64)
    for i, (scene, epoch, boxes, feats, quality) in enumerate(items):
        table[i] = (scene, epoch, len(boxes), (1 if feats is not None else 0) | (2 if quality is not None else 0))
    total = int(table[:, 2].sum()) if n else 0
    head = np.array([n, D, total, flags], np.int64)
    parts = [head.view(np.uint8), table.reshape(-1).view(np.uint8)]
    parts += [np.ascontiguousarray(it[2], abi.BOX_DTYPE).view(np.uint8).reshape(-1) for it in items]
    parts += [np.ascontiguousarray(it[4], np.float32).view(np.uint8).reshape(-1) for it in items if it[4] is not None]
    pre = sum(len(x) for x in parts)
    if feat_base:
        assert pre <= feat_base, f"prefix of {pre} B exceeds the agreed {feat_base} B (more scenes or rows than the capacity)"
        parts.append(np.zeros(feat_base - pre, np.uint8))
    parts += [np.ascontiguousarray(it[3], np.float32).view(np.uint8).reshape(-1) for it in items if it[3] is not None]
    return np.concatenate(parts) if parts else np.zeros(0, np.uint8)


def pack_share_into(dst: np.ndarray, items, D: int, feat_base: int = 0, flags: int = 0, bulk=None) -> int:
    """pack_share written straight into `dst` (a uint8 view of the pinned staging row of the rank): ONE copy of every array instead of
    a concatenation and a second copy into the staging buffer — at 64 scenes x 1000 x 512-d that is 131 MB per request set.  Returns the
    bytes written; the layout is pack_share's.  bulk: a list — the copies of the FEATURE rows are not made but appended to it as
    (destination view, source view) pairs, for the caller to run on several threads (numpy releases the GIL in them)."""
    from . import abi

    n = len(items)
    table = np.zeros((n, 4), np.int64)
    for i, (scene, epoch, boxes, feats, quality) in enumerate(items):
        table[i] = (scene, epoch, len(boxes), (1 if feats is not None else 0) | (2 if quality is not None else 0))
    total = int(table[:, 2].sum()) if n else 0
    o = 0

    def put(a: np.ndarray):
        nonlocal o
        v = a.reshape(-1).view(np.uint8)
        dst[o:o + len(v)] = v
        o += len(v)

    put(np.array([n, D, total, flags], np.int64))
    put(table)
    for it in items:
        put(np.ascontiguousarray(it[2], abi.BOX_DTYPE))
    for it in items:
        if it[4] is not None:
            put(np.ascontiguousarray(it[4], np.float32))
    if feat_base:
        assert o <= feat_base, f"prefix of {o} B exceeds the agreed {feat_base} B (more scenes or rows than the capacity)"
        dst[o:feat_base] = 0
        o = feat_base
    for it in items:
        if it[3] is not None:
            src = np.ascontiguousarray(it[3], np.float32).reshape(-1).view(np.uint8)
            if bulk is None:
                dst[o:o + len(src)] = src
            else:
                bulk.append((dst[o:o + len(src)], src))
            o += len(src)
    return o


def unpack_share(buf: np.ndarray, feat_base: int = 0, bulk=None):
    """Inverse of pack_share: views into `buf` (no copies).  bulk = None: the feature rows are views of buf as well; bulk = an
    integer ADDRESS: buf holds the prefix only and the rows are returned as addresses bulk + offset (device memory)."""
    from . import abi

    n, D, total, flags = (int(x) for x in buf[:32].view(np.int64))
    o = 32
    table = buf[o:o + 32 * n].view(np.int64).reshape(n, 4)
    o += 32 * n
    items = []
    counts = [int(c) for c in table[:, 2]]
    boxes = []
    for c in counts:
        boxes.append(buf[o:o + c * abi.BOX_DTYPE.itemsize].view(abi.BOX_DTYPE))
        o += c * abi.BOX_DTYPE.itemsize
    quality = []
    for i, c in enumerate(counts):
        if table[i, 3] & 2:
            quality.append(buf[o:o + 4 * c].view(np.float32))
            o += 4 * c
        else:
            quality.append(None)
    if feat_base:
        o = feat_base
    for i, c in enumerate(counts):
        ft = None
        if table[i, 3] & 1:
            ft = buf[o:o + 4 * c * D].view(np.float32).reshape(c, D) if bulk is None else int(bulk) + o
            o += 4 * c * D
        items.append((int(table[i, 0]), int(table[i, 1]), boxes[i], ft, quality[i]))
    return items, flags


class ShardedAssociator:
    """SPMD wrapper around one Engine per rank.  Root: associate(request set) -> [(ids, votes)] in request order; other ranks:
    serve_forever() (or associate(None) in lockstep).  `capacity_bytes` bounds the FEATURE bytes of one rank's share of a request
    set, `capacity_rows` its detections, `max_scenes` its scenes.  Tracks are upserted per scene on the owning rank (upsert is a
    collective too: the root passes the arrays).  A request set that exceeds a capacity is refused on EVERY rank (the root validates
    before any collective and sends the verdict with the shares): no rank is left waiting in a collective."""

    def __init__(self, engine, capacity_bytes: int, capacity_rows: int, group=None, root: int = 0, device=None, max_scenes: int = 64):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.eng, self.group, self.root = engine, group, root
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        self.device = device
        self.max_scenes = int(max_scenes)
        self.cap_r = int(capacity_rows)
        self.feat_base = prefix_bytes(self.max_scenes, self.cap_r)
        self.cap_b = self.feat_base + (int(capacity_bytes) + 255) // 256 * 256
        pin = device.type == "cuda"
        self.h_req = torch.zeros(self.feat_base if pin else self.cap_b, dtype=torch.uint8, pin_memory=pin)  # GPUs: the prefix only
        self.h_res = torch.zeros(self.cap_r * 9 + 8, dtype=torch.uint8, pin_memory=pin)   # ids[rows] (u64), votes[rows], then one status byte (1 = this rank failed)
        self.d_req = torch.zeros(self.cap_b, dtype=torch.uint8, device=device)
        self.d_res = torch.zeros(self.cap_r * 9 + 8, dtype=torch.uint8, device=device)
        # the bulk of a share is read where the collective put it: the receive buffer is a registered device block of the engine
        self.in_place = pin and hasattr(engine, "register_device_block")
        if self.in_place:
            engine.register_device_block(self.d_req.data_ptr(), self.cap_b, device.index if device.index is not None else torch.cuda.current_device())
        if self.rank == root:
            self.h_all = torch.zeros((self.world, self.cap_b), dtype=torch.uint8, pin_memory=pin)
            self.d_all = [torch.zeros(self.cap_b, dtype=torch.uint8, device=device) for _ in range(self.world)] if self.world > 1 else None
            self.d_gather = [torch.zeros(self.cap_r * 9 + 8, dtype=torch.uint8, device=device) for _ in range(self.world)]
        self.last_local_ms = 0.0
        self._pool = None

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        if self.in_place:
            self.eng.unregister_device_block(self.d_req.data_ptr())
            self.in_place = False

    def _run_share(self, buf: np.ndarray, bulk=None):
        """This rank's share -> its engine (one sa_associate_batch) -> ids | votes packed into h_res.  Returns False on shutdown."""
        import time

        from . import abi
        from .engine import Engine

        items, flags = unpack_share(buf, self.feat_base, bulk)
        if flags & FLAG_SHUTDOWN:
            return False
        if flags & FLAG_ABORT:
            raise RequestRefused("the root refused this request set (a rank's share exceeds the capacities agreed at construction)")
        t0 = time.perf_counter()
        dets = []
        for (_, _, b, f, q) in items:
            if isinstance(f, int):   # device address inside the registered receive buffer
                dets.append(abi.make_detections(b, feats_device_ptr=f, feat_quality=q))
            else:
                dets.append(abi.make_detections(b, feats=f, feat_quality=q))
        req, res, outs = Engine.make_requests([(it[0], it[1], d) for it, d in zip(items, dets)])
        if items:
            self.eng.associate_batch(req, res)
        self.last_local_ms = 1e3 * (time.perf_counter() - t0)
        total = sum(len(o[0]) for o in outs)
        out = self.h_res.numpy()
        if total:
            out[: 8 * total] = np.concatenate([o[0] for o in outs]).view(np.uint8)
            out[8 * self.cap_r: 8 * self.cap_r + total] = np.concatenate([o[1] for o in outs])
        return True

    def associate(self, items=None, shutdown: bool = False):
        torch, dist = self.torch, self.dist
        is_root = self.rank == self.root
        refused = None
        if is_root:
            shares = [[] for _ in range(self.world)]
            where = []
            if not shutdown:
                for it in items:
                    r = owner(it[0], self.world)
                    where.append((r, len(shares[r])))
                    shares[r].append(it)
            D = self.eng.cfg.feature_len if self.eng.cfg is not None else 0
            # every capacity is checked HERE, before the first collective: a refusal travels with the shares
            for r in range(self.world):
                rows = sum(len(it[2]) for it in shares[r])
                fbytes = sum(4 * D * len(it[2]) for it in shares[r] if it[3] is not None)
                if rows > self.cap_r:
                    refused = f"rank {r}'s share holds {rows} detections, capacity_rows is {self.cap_r}"
                elif len(shares[r]) > self.max_scenes:
                    refused = f"rank {r}'s share holds {len(shares[r])} scenes, max_scenes is {self.max_scenes}"
                elif self.feat_base + fbytes > self.cap_b:
                    refused = f"rank {r}'s share holds {fbytes} B of features, capacity_bytes is {self.cap_b - self.feat_base}"
            ha = self.h_all.numpy()
            bulk = []
            for r in range(self.world):
                flags = FLAG_SHUTDOWN if shutdown else (FLAG_ABORT if refused else 0)
                pack_share_into(ha[r], [] if (shutdown or refused) else shares[r], D, self.feat_base, flags, bulk)
            # the feature rows (megabytes per scene) on a few threads: one host core copies ~10 GB/s, the ingest point has more
            if len(bulk) > 1 and sum(len(d) for d, _ in bulk) > (8 << 20):
                if self._pool is None:
                    from concurrent.futures import ThreadPoolExecutor

                    self._pool = ThreadPoolExecutor(max_workers=8)
                list(self._pool.map(lambda p: np.copyto(p[0], p[1]), bulk))
            else:
                for d_, s_ in bulk:
                    np.copyto(d_, s_)
        if self.world > 1:
            if is_root:
                for r in range(self.world):
                    self.d_all[r].copy_(self.h_all[r], non_blocking=True)
            dist.scatter(self.d_req, self.d_all if is_root else None, src=self.root, group=self.group)
        elif self.in_place:
            self.d_req.copy_(self.h_all[0], non_blocking=True)   # one rank: the share still goes where the engine will read it
        if is_root and refused:   # (the other ranks learn it from their share's flags and skip the set)
            raise RequestRefused("request set refused: " + refused)
        # Between the scatter and the gather every rank MUST reach the gather, whatever its engine does: a rank that raised here and
        # went back to its scatter would leave the root blocked in the gather (or the collectives out of step).  A failure is carried
        # to the root as the status byte of the rank's result block and re-raised on the failing rank after the gather.
        failure = None
        alive = True
        self.h_res.numpy()[9 * self.cap_r] = 0
        try:
            if self.in_place:
                self.h_req.copy_(self.d_req[: self.feat_base])        # the prefix only; the feature rows stay in device memory
                alive = self._run_share(self.h_req.numpy(), bulk=self.d_req.data_ptr())
            elif self.world > 1:
                self.h_req.copy_(self.d_req)
                alive = self._run_share(self.h_req.numpy())
            else:
                alive = self._run_share(self.h_all.numpy()[0])
        except RequestRefused:
            raise   # (every rank sees the same flag: nobody enters the gather)
        except Exception as ex:  # noqa: BLE001 - anything the engine / the unpacking raised
            failure = ex
            self.h_res.numpy()[9 * self.cap_r] = 1
        if not alive:
            return None
        if self.world > 1:
            self.d_res.copy_(self.h_res, non_blocking=True)
            dist.gather(self.d_res, self.d_gather if is_root else None, dst=self.root, group=self.group)
        if failure is not None:
            raise failure
        if not is_root:
            return ()
        out = []
        per_rank = [self.d_gather[r].cpu().numpy() if self.world > 1 else self.h_res.numpy() for r in range(self.world)]
        bad = [r for r in range(self.world) if per_rank[r][9 * self.cap_r]]
        if bad:
            raise ShardFailed(f"rank(s) {bad} failed inside their share of the request set (their own exception is raised there)")
        offs = [[0] for _ in range(self.world)]
        for r in range(self.world):
            for it in shares[r]:
                offs[r].append(offs[r][-1] + len(it[2]))
        for (r, k) in where:
            a, b = offs[r][k], offs[r][k + 1]
            out.append((per_rank[r][8 * a: 8 * b].view(np.uint64).copy(), per_rank[r][8 * self.cap_r + a: 8 * self.cap_r + b].copy()))
        return out

    def upsert_arrays(self, scene_id: int, **arrays):
        """Collective: arrays as for abi.make_tracks (ids, boxes, epochs, kf_mean, kf_cov, feats, feat_present); given on the root,
        None elsewhere."""
        from . import abi

        obj = [arrays if self.rank == self.root else None]
        if self.world > 1:
            self.dist.broadcast_object_list(obj, src=self.root, group=self.group)
        if owner(scene_id, self.world) == self.rank:
            self.eng.upsert(scene_id, abi.make_tracks(**obj[0]))

    def serve_forever(self):
        """Worker loop.  A REFUSED request set (RequestRefused: decided by the root before the first collective, seen by every rank)
        is skipped; anything else — an engine error inside this rank's share (reported to the root through the gather first), a dead
        process group — ends the loop by raising."""
        assert self.rank != self.root
        while True:
            try:
                if self.associate(None) is None:
                    return
            except RequestRefused:
                continue

    def shutdown(self):
        assert self.rank == self.root
        self.associate(None, shutdown=True)
        self.close()

def cpu_share(local_rank: int, local_world: int, allowed=None, siblings_of=None):
    """The CPUs of rank `local_rank` of `local_world` processes on one host: the host's PHYSICAL cores (a core = its hardware threads, read
    from /sys/devices/system/cpu/cpuN/topology), in (package, first thread) order, are dealt out in contiguous runs — low ranks on the
    first socket, high ranks on the last, the way the GPUs of an 8-GPU node hang off the sockets — and a rank gets every hardware thread of
    its cores that the process may run on.  Without this the ranks' threads land wherever the scheduler likes: a rank's pinned staging
    rows on the far socket's memory, and the facade's pool workers (sa_pool.h binds them to the CPUs next to the caller's INSIDE the
    process's allowed set) of two ranks on the same cores.  Returns a sorted list (empty: leave the process alone)."""
    import os

    if allowed is None:
        allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
    if local_world <= 1 or not allowed:
        return []
    if siblings_of is None:
        def siblings_of(c):
            pkg, sib = 0, [c]
            try:
                base = f"/sys/devices/system/cpu/cpu{c}/topology/"
                with open(base + "physical_package_id") as f:
                    pkg = int(f.read())
                with open(base + "thread_siblings_list") as f:
                    sib = []
                    for part in f.read().strip().split(","):
                        lo, _, hi = part.partition("-")
                        sib.extend(range(int(lo), int(hi or lo) + 1))
            except (OSError, ValueError):
                pass
            return pkg, sib
    ok = set(allowed)
    cores = {}
    for c in allowed:
        pkg, sib = siblings_of(c)
        sib = tuple(sorted(x for x in sib if x in ok)) or (c,)
        cores.setdefault((pkg, sib[0]), sib)
    order = [cores[k] for k in sorted(cores)]
    if len(order) < local_world:
        return []
    lo = (len(order) * local_rank) // local_world
    hi = (len(order) * (local_rank + 1)) // local_world
    return sorted(c for sib in order[lo:hi] for c in sib)


def bind_rank_to_cpu_share(local_rank: int, local_world: int):
    """os.sched_setaffinity to cpu_share(); never fatal (a container may forbid it).  Returns the CPUs bound to, or []."""
    import os

    try:
        cpus = cpu_share(local_rank, local_world)
        if cpus:
            os.sched_setaffinity(0, cpus)
        return cpus
    except (OSError, ValueError, AttributeError):
        return []


class ResultGather:
    """The ONLY exchange of a multi-GPU run whose ranks ingest their own scenes (a camera's detector feeds the GPU that owns the camera's
    scene: `scene_id % world`): every step each rank hands over the ids and vote types of its scenes, the root receives them — KB-scale,
    `dist.gather` (RCCL over xGMI on the GPU box; gloo in the CPU tests).  Up to `depth` steps are in flight (each with buffers of its own),
    so the gather of step n rides beside the kernels of step n + 1.  Nothing of the request travels: the reference's fan-out of scenes to
    voting threads (sort/batch_api.rs:197-207, 278-288) becomes "the scene lives where its detections arrive".

    Wire: ids[rows] (u64) | votes[rows] (u8), rows = this rank's detections in scene order; `capacity_rows` bounds them on every rank."""

    def __init__(self, capacity_rows: int, group=None, root: int = 0, device=None, depth: int = 3, loopback: bool = False):
        """loopback: issue the gather in a group of ONE rank as well (a test of the backend's path on a single GPU)."""
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group, self.root = torch, dist, group, root
        self.loopback = bool(loopback)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        self.device, self.cap = device, int(capacity_rows)
        pin = device.type == "cuda"
        nb = 9 * self.cap
        self.h = [torch.zeros(nb, dtype=torch.uint8, pin_memory=pin) for _ in range(depth)]
        self.d = [torch.zeros(nb, dtype=torch.uint8, device=device) for _ in range(depth)] if pin else self.h
        self.recv = ([[torch.zeros(nb, dtype=torch.uint8, device=device) for _ in range(self.world)] for _ in range(depth)]
                     if self.rank == root else [None] * depth)
        self.work = [None] * depth
        # (GPU backends: Work.wait() orders STREAMS, not the host — the host may only rewrite a pinned staging row once the copy that
        # read it has actually run: an event behind every copy, synchronised before the row's next use)
        self.copied = [torch.cuda.Event() for _ in range(depth)] if pin else None
        self.rows = [0] * depth
        self.at = 0
        self.steps = 0

    def push(self, outs):
        """outs: [(ids u64[n], votes u8[n])] of this rank's scenes, in scene order.  Returns at once; the buffers of the step `depth` pushes
        ago are waited for first."""
        k = self.at
        if self.work[k] is not None:
            self.work[k].wait()
            self.work[k] = None
            if self.copied is not None:
                self.copied[k].synchronize()
        buf = self.h[k].numpy()
        o = 0
        for ids, votes in outs:
            n = len(ids)
            if o + n > self.cap:
                raise ValueError(f"a rank's share holds more than capacity_rows = {self.cap} detections")
            buf[8 * o: 8 * (o + n)] = np.ascontiguousarray(ids, np.uint64).view(np.uint8)
            buf[8 * self.cap + o: 8 * self.cap + o + n] = votes
            o += n
        self.rows[k] = o
        if self.d is not self.h:
            self.d[k].copy_(self.h[k], non_blocking=True)
            self.copied[k].record()
        if self.world > 1 or self.loopback:
            self.work[k] = self.dist.gather(self.d[k], self.recv[k], dst=self.root, group=self.group, async_op=True)
        self.at = (k + 1) % len(self.h)
        self.steps += 1

    def drain(self):
        for k in range(len(self.work)):
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None
        if self.device.type == "cuda":
            self.torch.cuda.synchronize()

    def last(self, rows_per_rank):
        """Root, after drain(): the last step's (ids, votes) of every rank — rows_per_rank[r] detections each."""
        assert self.rank == self.root
        k = (self.at - 1) % len(self.h)
        out = []
        for r in range(self.world):
            raw = (self.recv[k][r] if self.world > 1 or self.loopback else self.d[k]).cpu().numpy()
            n = int(rows_per_rank[r])
            out.append((raw[: 8 * n].view(np.uint64).copy(), raw[8 * self.cap: 8 * self.cap + n].copy()))
        return out
