"""Seeded synthetic inputs for the association path.  numpy only — used by tests and bench.py.

Distributions follow the reference's own generators: BoxGen2 / FeatGen (src/examples.rs:188-249, 266-293) and the
bench layouts (benches/simple_sort_iou_tracker.rs:29-61, benches/simple_visual_sort_tracker.rs:135-141), plus the
"dense" layouts SURVEY §8d defines so that cost cells are actually evaluated instead of pruned."""
from __future__ import annotations

import numpy as np

from . import abi


def dense_boxes(rng, n, canvas=(1920.0, 1080.0), oriented=False, h=(40.0, 200.0), aspect=(0.3, 0.6), conf=(0.3, 1.0)):
    """Boxes with centres U(canvas), height U(h), aspect U(aspect), confidence U(conf); angle U(0,1) if oriented
    (benches/simple_sort_iou_tracker_oriented.rs:72)."""
    return abi.make_boxes(
        rng.uniform(0, canvas[0], n), rng.uniform(0, canvas[1], n), rng.uniform(*aspect, n), rng.uniform(*h, n),
        confidence=rng.uniform(*conf, n), angle=rng.uniform(0.0, 1.0, n) if oriented else None,
    )


def diagonal_boxes(n, step=1000.0, w=50.0, hgt=50.0):
    """The reference bench layout: object i at (step*i, step*i), 50x50 (benches/simple_sort_iou_tracker.rs:34-43)."""
    i = np.arange(n, dtype=np.float32)
    return abi.ltwh(step * i, step * i, np.full(n, w, np.float32), np.full(n, hgt, np.float32))


def jitter_boxes(rng, boxes, pos_sigma=2.0, size_rel=0.001, angle_sigma=0.0, conf=(0.3, 1.0)):
    """Next-frame detections of the same objects: centre + N(0, sigma px), size * (1 + U(+-size_rel))."""
    n = len(boxes)
    out = boxes.copy()
    out["xc"] += rng.normal(0, pos_sigma, n).astype(np.float32)
    out["yc"] += rng.normal(0, pos_sigma, n).astype(np.float32)
    out["height"] *= (1.0 + rng.uniform(-size_rel, size_rel, n)).astype(np.float32)
    out["aspect"] *= (1.0 + rng.uniform(-size_rel, size_rel, n)).astype(np.float32)
    if angle_sigma > 0:
        out["angle"] += rng.normal(0, angle_sigma, n).astype(np.float32)
    out["confidence"] = rng.uniform(*conf, n).astype(np.float32)
    return out


def reid_identities(rng, n, d):
    """Unit-normalised |N(0,1)| vectors: non-negative like the ReID features of python/bugfixes/bug_vs_1/in/*.json."""
    f = np.abs(rng.standard_normal((n, d))).astype(np.float32)
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    return f.astype(np.float32)


def observe(rng, identities, jitter=0.01):
    """One observation per identity: identity + U(-jitter, jitter) per component (FeatGen, examples.rs:283-292)."""
    return (identities + rng.uniform(-jitter, jitter, identities.shape).astype(np.float32)).astype(np.float32)


def visual_scene(rng, n_tracks, n_dets, d, k, canvas=(1920.0, 1080.0), oriented=False, new_fraction=0.0,
                 feat_jitter=0.01, pos_sigma=2.0):
    """A VisualSORT scene-frame: T stored tracks with a K-deep feature bank, N detections of (mostly) the same
    objects in shuffled order.  Returns a dict of numpy arrays."""
    n_ident = max(n_tracks, n_dets)
    ident = reid_identities(rng, n_ident, d)
    tboxes = dense_boxes(rng, n_tracks, canvas, oriented)
    bank = np.stack([observe(rng, ident[:n_tracks], feat_jitter) for _ in range(k)], axis=1)  # T x K x D
    present = np.ones((n_tracks, k), np.uint8)
    # detections: objects perm[:n_dets]; objects beyond the track table are new
    perm = rng.permutation(n_ident)[:n_dets]
    dboxes = dense_boxes(rng, n_dets, canvas, oriented)
    old = perm < n_tracks
    jb = jitter_boxes(rng, tboxes, pos_sigma)
    dboxes[old] = jb[perm[old]]
    if new_fraction > 0:
        fresh = rng.uniform(size=n_dets) < new_fraction
        repl = dense_boxes(rng, n_dets, canvas, oriented)
        dboxes[fresh] = repl[fresh]
        perm = perm.copy()
        perm[fresh] = -1
    dfeat = observe(rng, ident[np.where(perm >= 0, perm, 0)], feat_jitter)
    if new_fraction > 0:
        nf = reid_identities(rng, n_dets, d)
        dfeat[perm < 0] = nf[perm < 0]
    truth = np.where((perm >= 0) & (perm < n_tracks), perm + 1, 0).astype(np.uint64)  # track ids are slot+1
    return dict(
        track_ids=np.arange(1, n_tracks + 1, dtype=np.uint64), track_boxes=tboxes, track_epochs=np.zeros(n_tracks, np.uint64),
        track_feats=bank, track_present=present, det_boxes=dboxes, det_feats=dfeat,
        det_quality=rng.uniform(0.5, 1.0, n_dets).astype(np.float32), truth=truth,
    )


def sort_scene(rng, n_tracks, n_dets, canvas=(4096.0, 4096.0), oriented=False, pos_sigma=2.0):
    """A SORT scene-frame: detections = shuffled tracks + jitter (+ random extras when n_dets > n_tracks)."""
    tboxes = dense_boxes(rng, n_tracks, canvas, oriented)
    n_ident = max(n_tracks, n_dets)
    perm = rng.permutation(n_ident)[:n_dets]
    dboxes = dense_boxes(rng, n_dets, canvas, oriented)
    old = perm < n_tracks
    jb = jitter_boxes(rng, tboxes, pos_sigma, angle_sigma=0.01 if oriented else 0.0)
    dboxes[old] = jb[perm[old]]
    truth = np.where(old, perm + 1, 0).astype(np.uint64)
    return dict(track_ids=np.arange(1, n_tracks + 1, dtype=np.uint64), track_boxes=tboxes,
                track_epochs=np.zeros(n_tracks, np.uint64), det_boxes=dboxes, truth=truth)
