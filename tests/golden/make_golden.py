#!/usr/bin/env python
"""Regenerates tests/golden/*.npz.  Run in the BUILD container (where /root/reference is mounted):

    python tests/golden/make_golden.py

What goes in:
  * the INPUT data of the reference's regression fixtures for this path, read from the reference tree —
        python/bugfixes/github-84.py:8-148        BOXES_1 / BOXES_2, 2 frames x 23 oriented boxes, Sort(IoU 0.3)
        python/bugfixes/bug_vs_1/in/*.json        2 + 2 frames x 3 detections with 512-d ReID features,
        python/bugfixes/bug_vs_1/bug_visual_sort.py:19-36   VisualSort options of that fixture
    (data only, stored as float arrays; no reference source is copied),
  * seeded synthetic scene-frames (similari_amd/synth.py) at sizes the oracle finishes instantly,
  * for every input, the outputs of the CPU oracle (oracle/liboracle.so).

The reference is a Rust crate that cannot be built in this image and ships NO expected outputs for these fixtures (the
scripts only assert "no panic" and "track ids unique per frame"), so the expected values are the oracle's, and the oracle
itself is pinned on the reference's literal known-answer tests in tests/test_oracle_kat.py.  tests/test_golden.py checks
(cpu) that the oracle still reproduces these files bit for bit and that the reference's invariants hold, and (gpu) that
the HIP path reproduces them through the C ABI.  The GPU box has no /root/reference: only the .npz files travel."""
from __future__ import annotations

import ast
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import oracle_lib as O  # noqa: E402
from similari_amd import abi, synth  # noqa: E402
from similari_amd import trackers as TR  # noqa: E402

REF = Path("/root/reference/python/bugfixes")


def track_rows(tracks):
    """SortTrack list -> float64 [n, 16]: id, epoch, scene, length, vote, custom(or -1), predicted(5: xc yc angle|nan aspect h),
    observed(5)."""
    out = np.zeros((len(tracks), 16), np.float64)
    for i, t in enumerate(tracks):
        def b5(b):
            return [b.xc, b.yc, np.nan if b.angle is None else b.angle, b.aspect, b.height]
        out[i] = [t.id, t.epoch, t.scene_id, t.length, t.voting_type, -1 if t.custom_object_id is None else t.custom_object_id,
                  *b5(t.predicted_bbox), *b5(t.observed_bbox)]
    return out


def github84_inputs():
    tree = ast.parse((REF / "github-84.py").read_text())
    vals = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") in ("BOXES_1", "BOXES_2"):
            vals[node.targets[0].id] = np.asarray(ast.literal_eval(node.value), np.float64)
    return vals["BOXES_1"], vals["BOXES_2"]


def github84_tracker(backend_cls):
    o, keep = TR.sort_options(10, 5, TR.PositionalMetricType.iou(0.3), 0.05, None, 1.0 / 20.0, 1.0 / 160.0)
    return backend_cls(o, keep)


def u2d_list(rows):
    return [(TR.Universal2DBox(float(r[0]), float(r[1]), float(r[2]), float(r[3]), float(r[4]), 1.0), None) for r in rows]


def make_github84():
    b1, b2 = github84_inputs()
    trk = github84_tracker(O.OracleTracker)
    f1 = track_rows(trk.predict(u2d_list(b1)))
    f2 = track_rows(trk.predict(u2d_list(b2)))
    trk.close()
    # the association of frame 2 in isolation: stored tracks = frame-1 predicted boxes, epoch 1 -> candidates at epoch 2
    cfg = abi.make_config(positional="iou", positional_threshold=0.3, positional_min_confidence=0.05, max_idle_epochs=5)
    tb = abi.make_boxes(f1[:, 6], f1[:, 7], f1[:, 9], f1[:, 10], confidence=np.ones(len(f1)), angle=f1[:, 8])
    tracks = abi.make_tracks(f1[:, 0].astype(np.uint64), tb, f1[:, 1].astype(np.uint64))
    db = abi.make_boxes(b2[:, 0], b2[:, 1], b2[:, 3], b2[:, 4], confidence=np.ones(len(b2)), angle=b2[:, 2])
    ref = O.associate(cfg, tracks, 2, abi.make_detections(db))
    np.savez_compressed(HERE / "github84.npz", boxes_1=b1, boxes_2=b2, frame_1=f1, frame_2=f2, assoc_positional=ref["positional"],
                        assoc_quantised=ref["quantised"], assoc_track_id=ref["track_id"], assoc_voting_type=ref["voting_type"],
                        assoc_total_weight=np.int64(ref["total_weight"]))
    print("github84:", len(b1), len(b2), "frame-2 matches", int((ref["track_id"] != 0).sum()))


def bug_vs_1_options():
    c = TR.SpatioTemporalConstraints().add_constraints([(1, 1.0)])
    return (TR.VisualSortOptions().spatio_temporal_constraints(c).max_idle_epochs(3).kept_history_length(10)
            .visual_metric(TR.VisualSortMetricType.euclidean(1.0)).positional_metric(TR.PositionalMetricType.maha())
            .visual_minimal_track_length(1).visual_minimal_area(5.0).visual_minimal_quality_use(0.45)
            .visual_minimal_quality_collect(0.5).visual_max_observations(5).visual_min_votes(1))


def load_frame(path):
    objs = json.loads(Path(path).read_text())
    boxes = np.array([[o["bbox"]["xc"], o["bbox"]["yc"], np.nan if o["bbox"]["angle"] is None else o["bbox"]["angle"],
                       o["bbox"]["aspect"], o["bbox"]["height"], o["bbox"]["confidence"]] for o in objs], np.float64)
    feats = np.array([o["feature"] for o in objs], np.float32)
    quality = np.array([o["feature_quality"] for o in objs], np.float32)
    return boxes, feats, quality


def observations(boxes, feats, quality):
    out = []
    for b, f, q in zip(boxes, feats, quality):
        bx = TR.Universal2DBox(float(b[0]), float(b[1]), None if np.isnan(b[2]) else float(b[2]), float(b[3]), float(b[4]), float(b[5]))
        out.append(TR.VisualSortObservation(f, float(q), bx, None))
    return out


def make_bug_vs_1():
    seqs = {"in": [REF / "bug_vs_1/in/in-1.json", REF / "bug_vs_1/in/in-2.json"],
            "fixed": [REF / "bug_vs_1/in/fixed-1/bug_vs_1.json", REF / "bug_vs_1/in/fixed-1/bug_vs_2.json"]}
    save = {}
    for name, files in seqs.items():
        o, keep = TR.visual_options(bug_vs_1_options(), 512)
        trk = O.OracleTracker(o, keep)
        for k, f in enumerate(files):
            boxes, feats, quality = load_frame(f)
            res = track_rows(trk.predict(observations(boxes, feats, quality)))
            save[f"{name}_{k}_boxes"] = boxes
            save[f"{name}_{k}_feats"] = feats
            save[f"{name}_{k}_quality"] = quality
            save[f"{name}_{k}_tracks"] = res
            print("bug_vs_1", name, k, "ids", res[:, 0].astype(int).tolist(), "votes", res[:, 4].astype(int).tolist())
        trk.close()
    np.savez_compressed(HERE / "bug_vs_1.npz", **save)


SYNTH_CASES = {
    # name: (kind, seed, T, N, D, K, config kwargs)
    "visual_cosine_k2": ("visual", 101, 50, 40, 64, 2, dict(positional="iou", positional_threshold=0.3, visual="cosine", visual_threshold=0.2,
                                                            visual_min_votes=1, visual_minimal_track_length=1, positional_min_confidence=0.1, max_idle_epochs=5)),
    "visual_euclid_k3": ("visual", 102, 33, 47, 36, 3, dict(positional="iou", positional_threshold=0.3, visual="euclidean", visual_threshold=0.3,
                                                            visual_min_votes=2, visual_minimal_track_length=2, positional_min_confidence=0.1, max_idle_epochs=5)),
    "sort_oriented_iou": ("sort", 103, 70, 64, 0, 1, dict(positional="iou", positional_threshold=0.3, max_idle_epochs=5)),
    "sort_iou_constraints": ("sort", 104, 60, 66, 0, 1, dict(positional="iou", positional_threshold=0.2, max_idle_epochs=4,
                                                             constraints=[(1, 0.05), (2, 0.5), (5, 1.5)])),
}


def synth_case(name):
    kind, seed, T, N, D, K, kw = SYNTH_CASES[name]
    rng = np.random.default_rng(seed)
    if kind == "visual":
        sc = synth.visual_scene(rng, T, N, D, K, canvas=(700.0, 500.0), new_fraction=0.1)
        cfg = abi.make_config(feature_len=D, max_observations=K, **kw)
    else:
        sc = synth.sort_scene(rng, T, N, canvas=(900.0, 700.0), oriented="oriented" in name)
        if "constraints" in name:
            sc["track_epochs"] = rng.integers(0, 9, T).astype(np.uint64)
        cfg = abi.make_config(**kw)
    return kind, cfg, sc


def case_io(kind, cfg, sc, epoch):
    kwt = dict(feats=sc["track_feats"], feat_present=sc["track_present"]) if kind == "visual" else {}
    kwd = dict(feats=sc["det_feats"], feat_quality=sc["det_quality"]) if kind == "visual" else {}
    tracks = abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"], **kwt)
    det = abi.make_detections(sc["det_boxes"], **kwd)
    return tracks, det


def make_synth():
    for name in SYNTH_CASES:
        kind, cfg, sc = synth_case(name)
        epoch = 8 if "constraints" in name else 1
        tracks, det = case_io(kind, cfg, sc, epoch)
        ref = O.associate(cfg, tracks, epoch, det)
        save = {k: v for k, v in sc.items() if isinstance(v, np.ndarray)}
        save.update({"out_" + k: np.asarray(v) for k, v in ref.items()})
        save["epoch"] = np.uint64(epoch)
        np.savez_compressed(HERE / f"synth_{name}.npz", **save)
        print("synth", name, "matched", int((ref["track_id"] != 0).sum()), "of", det.n)


if __name__ == "__main__":
    if not REF.exists():
        raise SystemExit("the reference tree is not mounted: golden fixtures can only be regenerated in the build container")
    make_github84()
    make_bug_vs_1()
    make_synth()
