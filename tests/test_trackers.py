"""Tracker-level tests.  The reference's own end-to-end tests (sort/simple_api.rs:280-432,
visual_sort/simple_api.rs:328-666) are ported literally and run against
  * the oracle's tracker loops (CPU, always) — this pins the oracle's orchestration on the reference's expectations,
  * the product facade on the GPU (marked gpu) — same assertions, so the parity tests read like the reference's.
Then the product is compared frame by frame with the oracle on seeded multi-frame sequences."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from similari_amd import abi, synth
from similari_amd import trackers as TR

IoU = TR.PositionalMetricType.iou
BB = TR.BoundingBox


def make(backend, kind, **kw):
    """kind: 'sort' | 'visual'; backend: 'oracle' | 'gpu' (host upkeep around the GPU association) | 'gpu_dev' (Kalman step,
    table refresh and feature bank on the GPU too: sa_tracks_apply)."""
    dev = backend == "gpu_dev"
    if kind == "sort":
        o, keep = TR.sort_options(kw.get("bbox_history", 10), kw.get("max_idle_epochs", 2), kw.get("method", IoU(0.3)),
                                  kw.get("min_confidence", 0.05), kw.get("constraints"), 1.0 / 20.0, 1.0 / 160.0,
                                  batch=kw.get("batch", False), device_upkeep=dev, workers=kw.get("workers", 0),
                                  devices=kw.get("devices"), spin_us=kw.get("spin_us", -1))
    else:
        o, keep = TR.visual_options(kw["opts"], kw["feature_len"], batch=kw.get("batch", False), device_upkeep=dev,
                                    workers=kw.get("workers", 0), devices=kw.get("devices"), spin_us=kw.get("spin_us", -1))
    if backend == "oracle":
        return O.OracleTracker(o, keep)
    return TR._Tracker(o, keep)


BACKENDS = [pytest.param("oracle", id="oracle"), pytest.param("gpu", id="gpu", marks=pytest.mark.gpu),
            pytest.param("gpu_dev", id="gpu_device_upkeep", marks=pytest.mark.gpu)]


def box_eq(a: TR.Universal2DBox, b: TR.Universal2DBox, eps=1e-5):
    # Universal2DBox PartialEq  bbox.rs:537-545
    return (abs(a.xc - b.xc) < eps and abs(a.yc - b.yc) < eps and ((a.angle or 0.0) - (b.angle or 0.0)) < eps
            and (a.aspect - b.aspect) < eps and (a.height - b.height) < eps)


# ---- sort/simple_api.rs:280-342 -------------------------------------------------------------------------
@pytest.mark.parametrize("backend", BACKENDS)
def test_sort(backend):
    t = make(backend, "sort", bbox_history=10, max_idle_epochs=2)
    assert t.current_epoch() == 0
    bb = BB(0.0, 0.0, 10.0, 20.0)
    v = t.predict([(bb.as_xyaah(), None)])
    assert t.wasted() == []
    assert len(v) == 1
    v = v[0]
    track_id = v.id
    assert v.custom_object_id is None and v.length == 1 and v.epoch == 1
    assert box_eq(v.observed_bbox, bb.as_xyaah())
    assert t.current_epoch() == 1

    bb = BB(0.1, 0.1, 10.1, 20.0)
    v = t.predict([(bb.as_xyaah(), 2)])
    assert t.wasted() == []
    v = v[0]
    assert v.custom_object_id == 2 and v.id == track_id and v.length == 2 and v.epoch == 2
    assert box_eq(v.observed_bbox, bb.as_xyaah())
    assert t.current_epoch() == 2

    bb = BB(10.1, 10.1, 10.1, 20.0)
    v = t.predict([(bb.as_xyaah(), 3)])
    assert len(v) == 1 and v[0].custom_object_id == 3 and v[0].id != track_id
    assert t.wasted() == []
    assert t.current_epoch() == 3

    assert t.predict([]) == []
    assert t.wasted() == []
    assert t.current_epoch() == 4

    assert t.predict([]) == []
    w = t.wasted()
    assert len(w) == 1 and w[0].id == track_id
    assert t.current_epoch() == 5
    t.close()


# ---- sort/simple_api.rs:344-372 -------------------------------------------------------------------------
@pytest.mark.parametrize("backend", BACKENDS)
def test_sort_with_scenes(backend):
    t = make(backend, "sort")
    bb = BB(0.0, 0.0, 10.0, 20.0).as_xyaah()
    assert t.current_epoch_with_scene(1) == 0 and t.current_epoch_with_scene(2) == 0
    t.predict_with_scene(1, [(bb, 4)])
    t.predict_with_scene(1, [(bb, 5)])
    assert t.current_epoch_with_scene(1) == 2 and t.current_epoch_with_scene(2) == 0
    t.predict_with_scene(2, [(bb, 6)])
    assert t.current_epoch_with_scene(1) == 2 and t.current_epoch_with_scene(2) == 1
    t.close()


# ---- sort/simple_api.rs:374-398 -------------------------------------------------------------------------
@pytest.mark.parametrize("backend", BACKENDS)
def test_idle_tracks(backend):
    t = make(backend, "sort")
    bb = BB(0.0, 0.0, 10.0, 20.0).as_xyaah()
    t.predict_with_scene(1, [(bb, 4)])
    assert t.idle_tracks_with_scene(1) == []
    t.predict_with_scene(1, [])
    idle = t.idle_tracks_with_scene(1)
    assert len(idle) == 1 and idle[0].id == 1
    t.close()


# ---- sort/simple_api.rs:400-432 -------------------------------------------------------------------------
@pytest.mark.parametrize("backend", BACKENDS)
def test_clear_wasted_tracks(backend):
    t = make(backend, "sort")
    bb = BB(0.0, 0.0, 10.0, 20.0).as_xyaah()
    t.predict_with_scene(1, [(bb, 4)])
    t.skip_epochs_for_scene(1, 3)
    assert t.wasted_count() == 1
    t.clear_wasted()
    assert t.wasted_count() == 0
    t.close()


# ---- visual_sort/simple_api.rs:328-666 (first track's life) ---------------------------------------------
def vobs(feat, q, bb, cid):
    return TR.VisualSortObservation(feat, q, bb.as_xyaah(), cid)


@pytest.mark.parametrize("backend", BACKENDS)
def test_visual_sort(backend):
    opts = (TR.VisualSortOptions().max_idle_epochs(3).kept_history_length(3)
            .visual_metric(TR.VisualSortMetricType.euclidean(1.0)).positional_metric(TR.PositionalMetricType.maha())
            .visual_minimal_track_length(2).visual_minimal_area(5.0).visual_minimal_quality_use(0.45)
            .visual_minimal_quality_collect(0.7).visual_max_observations(3).visual_min_votes(2))
    t = make(backend, "visual", opts=opts, feature_len=2)
    P, V = TR.VotingType.Positional, TR.VotingType.Visual

    tr = t.predict_with_scene(10, [vobs([1.0, 1.0], 0.9, BB(1.0, 1.0, 3.0, 5.0), 13)])[0]
    assert (tr.custom_object_id, tr.scene_id, tr.voting_type, tr.epoch) == (13, 10, P, 1)
    first = tr.id
    info = t.track_info(first)
    assert info == dict(visual_features_collected_count=1, observations=1, history=1, track_length=1)

    tr = t.predict_with_scene(1, [vobs([1.0, 1.0], 0.9, BB(1.0, 1.0, 3.0, 5.0), 133)])[0]  # another scene: new track
    assert (tr.custom_object_id, tr.scene_id, tr.voting_type, tr.epoch) == (133, 1, P, 1)
    assert tr.id != first

    tr = t.predict_with_scene(10, [vobs([0.95, 0.95], 0.93, BB(1.1, 1.1, 3.05, 5.01), 15)])[0]  # merge by position
    assert (tr.id, tr.custom_object_id, tr.voting_type, tr.epoch) == (first, 15, P, 2)
    info = t.track_info(first)
    assert (info["visual_features_collected_count"], info["track_length"], info["history"]) == (2, 2, 2)

    tr = t.predict_with_scene(10, [vobs(None, 0.93, BB(1.11, 1.15, 3.15, 5.05), 25)])[0]  # no feature
    assert (tr.id, tr.custom_object_id, tr.voting_type, tr.epoch) == (first, 25, P, 3)
    info = t.track_info(first)
    assert (info["visual_features_collected_count"], info["track_length"], info["history"]) == (2, 3, 3)

    tr = t.predict_with_scene(10, [vobs(None, 0.93, BB(1.15, 1.25, 3.10, 5.05), 2)])[0]
    assert (tr.id, tr.voting_type, tr.epoch) == (first, P, 4)
    info = t.track_info(first)
    assert (info["visual_features_collected_count"], info["track_length"], info["history"]) == (2, 4, 3)

    # feature of low quality: no use, no collect
    tr = t.predict_with_scene(10, [vobs([0.97, 0.97], 0.44, BB(1.15, 1.25, 3.10, 5.05), 2)])[0]
    assert (tr.id, tr.voting_type) == (first, P)
    info = t.track_info(first)
    assert (info["visual_features_collected_count"], info["track_length"]) == (2, 5)

    # use, but no collect
    tr = t.predict_with_scene(10, [vobs([0.97, 0.97], 0.6, BB(1.15, 1.25, 3.10, 5.05), 2)])[0]
    assert (tr.id, tr.voting_type) == (first, V)
    info = t.track_info(first)
    assert (info["visual_features_collected_count"], info["track_length"]) == (2, 6)

    # use and collect
    tr = t.predict_with_scene(10, [vobs([0.97, 0.97], 0.8, BB(1.15, 1.25, 3.10, 5.05), 2)])[0]
    assert (tr.id, tr.voting_type) == (first, V)
    info = t.track_info(first)
    assert (info["visual_features_collected_count"], info["track_length"], info["observations"]) == (3, 7, 3)

    # far away: new track
    tr = t.predict_with_scene(10, [vobs([0.1, 0.1], 0.9, BB(10.0, 10.0, 3.0, 5.0), 33)])[0]
    assert (tr.custom_object_id, tr.scene_id, tr.voting_type, tr.epoch) == (33, 10, P, 8)
    assert tr.id != first
    info = t.track_info(tr.id)
    assert info == dict(visual_features_collected_count=1, observations=1, history=1, track_length=1)
    t.close()


# ---- product vs oracle, frame by frame -------------------------------------------------------------------
def assert_tracks_equal(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert (x.id, x.epoch, x.scene_id, x.length, x.voting_type, x.custom_object_id) == \
               (y.id, y.epoch, y.scene_id, y.length, y.voting_type, y.custom_object_id)
        for bx, by in ((x.predicted_bbox, y.predicted_bbox), (x.observed_bbox, y.observed_bbox)):
            assert (bx.xc, bx.yc, bx.aspect, bx.height, bx.confidence) == (by.xc, by.yc, by.aspect, by.height, by.confidence)
            assert (bx.angle is None) == (by.angle is None)
            assert bx.angle is None or bx.angle == by.angle


def boxes_to_u2d(b):
    return [TR.Universal2DBox(float(r["xc"]), float(r["yc"]), float(r["angle"]) if r["has_angle"] else None,
                              float(r["aspect"]), float(r["height"]), float(r["confidence"])) for r in b]


def run_sort_sequence(method, oriented, seed, frames=8, n=60, scenes=(0,), batch=False, constraints=None, backend="gpu"):
    rng = np.random.default_rng(seed)
    kw = dict(bbox_history=3, max_idle_epochs=2, method=method, min_confidence=0.05, constraints=constraints, batch=batch)
    g, o = make(backend, "sort", **kw), make("oracle", "sort", **kw)
    try:
        world = {s: synth.dense_boxes(rng, n, (900.0, 700.0), oriented) for s in scenes}
        for f in range(frames):
            req = TR.PredictionBatchRequest()
            for s in scenes:
                world[s] = synth.jitter_boxes(rng, world[s], 2.0, angle_sigma=0.01 if oriented else 0.0)
                keep = rng.uniform(size=len(world[s])) > 0.15           # missed detections
                det = world[s][keep]
                extra = synth.dense_boxes(rng, int(rng.integers(0, 6)), (900.0, 700.0), oriented)  # false positives
                det = np.concatenate([det, extra])[rng.permutation(len(det) + len(extra))]
                for i, bx in enumerate(boxes_to_u2d(det)):
                    req.add(s, (bx, int(rng.integers(0, 1000)) if i % 3 == 0 else None))
                if f == 4 and len(world[s]) > 10:             # a few objects leave for good
                    world[s] = world[s][:-5]
            rg, ro = g.predict_batch(req), o.predict_batch(req)
            for s in scenes:
                assert_tracks_equal(rg[s], ro[s])
                ids = [x.id for x in rg[s]]
                assert len(ids) == len(set(ids)), "a track id repeats within one frame"
            if f % 3 == 2:
                assert sorted(x.id for x in g.wasted()) == sorted(x.id for x in o.wasted())
                for s in scenes:
                    assert_tracks_equal(sorted(g.idle_tracks_with_scene(s), key=lambda x: x.id),
                                        sorted(o.idle_tracks_with_scene(s), key=lambda x: x.id))
        assert g.active_tracks() == o.active_tracks()
        # Kalman states agree value for value
        for x in rg[scenes[0]][:10]:
            mg, cg = g.track_state(x.id)
            mo, co = o.track_state(x.id)
            np.testing.assert_array_equal(mg, mo)
            np.testing.assert_array_equal(cg, co)
    finally:
        g.close()
        o.close()


UPKEEP = pytest.mark.parametrize("backend", ["gpu", "gpu_dev"], ids=["host_upkeep", "device_upkeep"])


@pytest.mark.gpu
@UPKEEP
@pytest.mark.parametrize("oriented", [False, True])
def test_sort_iou_sequence_matches_oracle(oriented, backend):
    # (device upkeep + oriented boxes: the polygons take cos / sin from the host's libm, test_device_upkeep_keeps_oriented_polygons_bit_exact)
    run_sort_sequence(IoU(0.3), oriented, seed=11 + oriented, backend=backend)


@pytest.mark.gpu
def test_device_upkeep_keeps_oriented_polygons_bit_exact():
    """Oriented boxes under device-side upkeep, 220 frames: the polygon of every refreshed track row — what the next frame's IoU cells
    clip against — must be Polygon::from(&predicted box) with the HOST libm's cos / sin (bbox.rs:287-330; the reference's f64::cos /
    sin resolve to the same libm), bit for bit: the oracle's or_vertices of the predicted box the facade reports.  (Round 2 took the
    device's sincos there: one angle in a thousand differs in the last bit, and the bit-exact IoU / quantised gates inherit it.)
    The same frames through a host-upkeep facade (polygons from uploaded boxes) must give the same tracks, and its table the same
    polygons."""
    import ctypes as C

    rng = np.random.default_rng(123)
    kw = dict(bbox_history=3, max_idle_epochs=2, method=IoU(0.3), min_confidence=0.05)
    g, h = make("gpu_dev", "sort", **kw), make("gpu", "sort", **kw)
    L = O.lib()
    try:
        from similari_amd.engine import Engine
        eg = Engine.borrowed(g.lib, g.lib.sa_tracker_engine(g.h))
        eh = Engine.borrowed(h.lib, h.lib.sa_tracker_engine(h.h))
        world = synth.dense_boxes(rng, 70, (1000.0, 800.0), True)
        checked = 0
        for f in range(220):
            world = synth.jitter_boxes(rng, world, 2.0, angle_sigma=0.03)
            if f % 40 == 39:
                world["angle"] = rng.uniform(-3.1, 3.1, len(world)).astype(np.float32)   # every quadrant, large arguments too
            det = world[rng.uniform(size=len(world)) > 0.1]
            items = [(bx, None) for bx in boxes_to_u2d(det)]
            rg, rh = g.predict(items), h.predict(items)
            assert_tracks_equal(rg, rh)
            order = {int(t): r for r, t in enumerate(eg.order(0))}
            polys = eg.tap_track_polygons(0)
            np.testing.assert_array_equal(eg.order(0), eh.order(0))
            np.testing.assert_array_equal(polys, eh.tap_track_polygons(0))
            for x in rg:
                b = np.zeros(1, abi.BOX_DTYPE)
                pb = x.predicted_bbox
                b[0] = (pb.xc, pb.yc, pb.angle or 0.0, pb.aspect, pb.height, pb.confidence, 0 if pb.angle is None else 1, 0)
                ref = np.zeros(8, np.float64)
                L.or_vertices(O.box_ptr(b), ref.ctypes.data_as(C.POINTER(C.c_double)))
                np.testing.assert_array_equal(polys[order[x.id]].ravel(), ref, err_msg=f"frame {f} track {x.id}")
                checked += 1
        assert checked > 10000
    finally:
        g.close()
        h.close()


@pytest.mark.gpu
@UPKEEP
def test_sort_maha_sequence_matches_oracle(backend):
    run_sort_sequence(TR.PositionalMetricType.maha(), False, seed=13, backend=backend)


@pytest.mark.gpu
@UPKEEP
def test_batch_sort_scenes_and_constraints_match_oracle(backend):
    c = TR.SpatioTemporalConstraints().add_constraints([(1, 1.0), (2, 1.5)])
    run_sort_sequence(IoU(0.3), False, seed=17, scenes=(3, 7, 11), batch=True, constraints=c, n=40, backend=backend)


def run_visual_sequence(metric, positional, seed, frames=8, n=40, d=64, batch=False, backend="gpu", own=None, bank=3):
    rng = np.random.default_rng(seed)
    opts = (TR.VisualSortOptions().max_idle_epochs(2).kept_history_length(3).visual_metric(metric)
            .positional_metric(positional).visual_minimal_track_length(min(2, bank)).visual_minimal_area(500.0)
            .visual_minimal_quality_use(0.4).visual_minimal_quality_collect(0.6).visual_max_observations(bank).visual_min_votes(1))
    if own is not None:   # arm the own-area gates: the trackers compute exclusively_owned_areas shares themselves
        opts = opts.visual_minimal_own_area_percentage_use(own[0]).visual_minimal_own_area_percentage_collect(own[1])
    g = make(backend, "visual", opts=opts, feature_len=d, batch=batch)
    o = make("oracle", "visual", opts=opts, feature_len=d, batch=batch)
    try:
        ident = synth.reid_identities(rng, n, d)
        world = synth.dense_boxes(rng, n, (900.0, 700.0))
        for f in range(frames):
            world = synth.jitter_boxes(rng, world, 2.0)
            order = rng.permutation(n)
            keep = order[rng.uniform(size=n) > 0.1]
            feats = synth.observe(rng, ident[keep], 0.01)
            req = TR.PredictionBatchRequest()
            for k, (bx, ft) in enumerate(zip(boxes_to_u2d(world[keep]), feats)):
                q = float(rng.uniform(0.2, 1.0))
                req.add(5, TR.VisualSortObservation(None if k % 7 == 3 else ft, None if k % 5 == 0 else q, bx, k if k % 2 else None))
            rg, ro = g.predict_batch(req), o.predict_batch(req)
            assert_tracks_equal(rg[5], ro[5])
            for x in rg[5][:8]:
                assert g.track_info(x.id) == o.track_info(x.id)
        votes = [x.voting_type for x in rg[5]]
        assert TR.VotingType.Visual in votes
    finally:
        g.close()
        o.close()


@pytest.mark.gpu
@UPKEEP
def test_visual_cosine_sequence_matches_oracle(backend):
    run_visual_sequence(TR.VisualSortMetricType.cosine(0.5), IoU(0.3), seed=21, backend=backend)


@pytest.mark.gpu
@UPKEEP
def test_visual_cosine_single_observation_sequence_matches_oracle(backend):
    """One observation per track: the engine's contraction emits the BestFit partials itself (no weight matrix) and, with a
    feature length that is a multiple of 32, the first phase is the heterogeneous launch — the path of the default bench line."""
    run_visual_sequence(TR.VisualSortMetricType.cosine(0.5), IoU(0.3), seed=29, n=70, d=64, backend=backend, bank=1)


@pytest.mark.gpu
@UPKEEP
def test_visual_cosine_deepest_bank_sequence_matches_oracle(backend):
    """visual_max_observations = SA_MAX_BANK (16): k_apply_bank keeps the new observation's quality one slot past the bank."""
    run_visual_sequence(TR.VisualSortMetricType.cosine(0.5), IoU(0.3), seed=31, frames=20, n=24, d=32, backend=backend, bank=16)


@pytest.mark.gpu
@UPKEEP
def test_visual_euclid_maha_sequence_matches_oracle(backend):
    run_visual_sequence(TR.VisualSortMetricType.euclidean(0.5), TR.PositionalMetricType.maha(), seed=23, batch=True, backend=backend)


@pytest.mark.gpu
@UPKEEP
def test_visual_sequence_with_own_area_gates_matches_oracle(backend):
    """visual_sort/simple_api.rs:111-127: with either own-area threshold > 0 the tracker computes the shares of the frame's boxes
    (sa_own_areas on the device, or_own_area_shares in the oracle) and gates feature use / collection on them."""
    run_visual_sequence(TR.VisualSortMetricType.cosine(0.5), IoU(0.3), seed=27, n=60, backend=backend, own=(0.55, 0.7))


def test_oracle_own_area_gates_change_the_outcome():
    """The gates must actually bite on the sequence above, otherwise the parity test proves nothing."""
    def collected(own):
        rng = np.random.default_rng(27)
        opts = (TR.VisualSortOptions().max_idle_epochs(2).kept_history_length(3).visual_metric(TR.VisualSortMetricType.cosine(0.5))
                .positional_metric(IoU(0.3)).visual_minimal_track_length(2).visual_minimal_area(500.0)
                .visual_minimal_quality_use(0.4).visual_minimal_quality_collect(0.6).visual_max_observations(3).visual_min_votes(1))
        if own:
            opts = opts.visual_minimal_own_area_percentage_use(own[0]).visual_minimal_own_area_percentage_collect(own[1])
        o = make("oracle", "visual", opts=opts, feature_len=16)
        try:
            ident = synth.reid_identities(rng, 60, 16)
            world = synth.dense_boxes(rng, 60, (900.0, 700.0))
            tot = 0
            for f in range(4):
                world = synth.jitter_boxes(rng, world, 2.0)
                items = [TR.VisualSortObservation(ft, 0.9, bx, None) for bx, ft in zip(boxes_to_u2d(world), synth.observe(rng, ident, 0.01))]
                r = o.predict(items)
            for x in r:
                tot += o.track_info(x.id)["visual_features_collected_count"]
            return tot
        finally:
            o.close()
    assert collected((0.55, 0.7)) < collected(None)


@pytest.mark.gpu
@pytest.mark.parametrize("d", [48, 64], ids=["padded_rows", "lean_frames"])
def test_device_bank_equals_host_policy_bank(d):
    """The feature bank the device keeps (rows, presence flags, qualities, order) is what the host policy
    (optimize_observations, visual_sort/metric.rs:129-154) builds from the same observations: two facades, one with host
    upkeep (its bank is uploaded every frame), one with device upkeep, are fed the same frames and their engines' banks are
    read back and compared slot for slot.  d = 48: the rows are padded to 64 floats, the frame carries its preparation blocks for the
    bank step; d = 64: the frame runs lean and the bank step reads the uploaded rows and forms their norms itself."""
    import ctypes as C

    rng = np.random.default_rng(31)
    n = 30
    opts = (TR.VisualSortOptions().max_idle_epochs(3).kept_history_length(3).visual_metric(TR.VisualSortMetricType.cosine(0.5))
            .positional_metric(IoU(0.3)).visual_minimal_track_length(1).visual_minimal_area(500.0)
            .visual_minimal_quality_use(0.3).visual_minimal_quality_collect(0.5).visual_max_observations(3).visual_min_votes(1))
    h, g = make("gpu", "visual", opts=opts, feature_len=d), make("gpu_dev", "visual", opts=opts, feature_len=d)
    try:
        ident = synth.reid_identities(rng, n, d)
        world = synth.dense_boxes(rng, n, (900.0, 700.0))
        for f in range(7):
            world = synth.jitter_boxes(rng, world, 2.0)
            feats = synth.observe(rng, ident, 0.01)
            items = [TR.VisualSortObservation(None if (k + f) % 6 == 5 else ft, float(rng.uniform(0.2, 1.0)), bx, None)
                     for k, (bx, ft) in enumerate(zip(boxes_to_u2d(world), feats))]
            rh, rg = h.predict(items), g.predict(items)
            assert_tracks_equal(rg, rh)
        fp, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
        checked = 0
        for x in rg:
            banks = []
            for trk in (h, g):
                eng = trk.lib.sa_tracker_engine(trk.h)
                q, pres, ft = np.zeros(3, np.float32), np.zeros(3, np.uint8), np.zeros((3, d), np.float32)
                rc = trk.lib.sa_tracks_get_state(eng, 0, x.id, None, None, q.ctypes.data_as(fp), pres.ctypes.data_as(u8p), ft.ctypes.data_as(fp))
                assert rc == 0
                banks.append((q, pres, ft))
            (qh, ph, fh), (qg, pg, fg) = banks
            np.testing.assert_array_equal(ph, pg)
            np.testing.assert_array_equal(fh[ph != 0], fg[pg != 0])
            checked += int(pg.sum())
        assert checked > n
    finally:
        h.close()
        g.close()


@pytest.mark.gpu
@UPKEEP
@pytest.mark.parametrize("kind", ["sort", "visual"])
def test_churned_loop_with_eviction_matches_oracle(kind, backend):
    """Objects leave and enter every frame (max_idle_epochs 2, auto-waste only every 100th predict): the facade takes the tracks that
    can no longer match out of the ENGINE's table as soon as they are 64 and a sixteenth of it (sa_tracks_remove: the table closes ranks in
    order) while keeping them in its store — frame by frame the same tracks as the oracle's tracker, which keeps everything until
    auto_waste as the reference does; idle_tracks and wasted still see the evicted ones; the engine's table stays near the live set."""
    rng = np.random.default_rng(404 + (kind == "visual"))
    n, d = 220, 32
    if kind == "visual":
        # (cosine 0.9: random non-negative 32-d features sit at ~0.64 of each other — a lower threshold lets new objects revive idle tracks)
        opts = (TR.VisualSortOptions().max_idle_epochs(2).kept_history_length(3).visual_metric(TR.VisualSortMetricType.cosine(0.9))
                .positional_metric(IoU(0.3)).visual_minimal_track_length(1).visual_max_observations(2).visual_min_votes(1))
        g, o = make(backend, "visual", opts=opts, feature_len=d), make("oracle", "visual", opts=opts, feature_len=d)
    else:
        kw = dict(bbox_history=3, max_idle_epochs=2, method=IoU(0.3), min_confidence=0.05)
        g, o = make(backend, "sort", **kw), make("oracle", "sort", **kw)
    try:
        pool = n + 14 * 40
        world = synth.dense_boxes(rng, pool, (2600.0, 1800.0))
        ident = synth.reid_identities(rng, pool, d)
        active = np.arange(n)
        fresh = n
        rows_seen = []
        for f in range(14):
            world = synth.jitter_boxes(rng, world, 1.5)
            if f:
                gone = rng.choice(n, 40, replace=False)           # ~18 % of the objects leave, as many enter
                active[gone] = np.arange(fresh, fresh + 40)
                fresh += 40
            boxes = boxes_to_u2d(world[active])
            if kind == "visual":
                feats = synth.observe(rng, ident[active], 0.01)
                items = [TR.VisualSortObservation(feats[k], 0.9, boxes[k], None) for k in range(n)]
            else:
                items = [(boxes[k], None) for k in range(n)]
            rg, ro = g.predict(items), o.predict(items)
            assert_tracks_equal(rg, ro)
            cnt = C.c_uint32()
            g.lib.sa_tracks_count(g.lib.sa_tracker_engine(g.h), 0, C.byref(cnt))
            rows_seen.append(int(cnt.value))
            if f % 4 == 3:
                assert_tracks_equal(sorted(g.idle_tracks_with_scene(0), key=lambda x: x.id), sorted(o.idle_tracks_with_scene(0), key=lambda x: x.id))
        # eviction happened: the engine's table holds the live set + the recently idle tracks, not everything since frame 0
        assert g.active_tracks() == o.active_tracks()   # (the store: evicted tracks included, until they are wasted)
        assert rows_seen[-1] < n + 6 * 40, rows_seen
        assert max(rows_seen) > n, rows_seen
        assert sorted(x.id for x in g.wasted()) == sorted(x.id for x in o.wasted())
    finally:
        g.close()
        o.close()


@pytest.mark.gpu
@UPKEEP
@pytest.mark.parametrize("kind", ["sort", "visual"])
def test_loop_of_more_than_1024_objects_matches_oracle(kind, backend):
    """1300 objects (SORT) / 1100 objects (VisualSORT, one observation per track) with objects leaving and entering: frames of more than
    1024 detections against tables that grow towards 2048 rows — the one-workgroup tail with two rows and two columns per thread behind the
    facade (results, winning columns for the device upkeep, new-track ids), frame by frame the oracle tracker's tracks."""
    rng = np.random.default_rng(1300 + (kind == "visual"))
    n, d = (1300, 0) if kind == "sort" else (1100, 32)
    if kind == "visual":
        opts = (TR.VisualSortOptions().max_idle_epochs(2).kept_history_length(3).visual_metric(TR.VisualSortMetricType.cosine(0.9))
                .positional_metric(IoU(0.3)).visual_minimal_track_length(1).visual_max_observations(1).visual_min_votes(1))
        g, o = make(backend, "visual", opts=opts, feature_len=d), make("oracle", "visual", opts=opts, feature_len=d)
    else:
        kw = dict(bbox_history=3, max_idle_epochs=2, method=IoU(0.3), min_confidence=0.05)
        g, o = make(backend, "sort", **kw), make("oracle", "sort", **kw)
    try:
        pool = n + 6 * 120
        world = synth.dense_boxes(rng, pool, (6000.0, 4500.0))
        ident = synth.reid_identities(rng, pool, d) if kind == "visual" else None
        active = np.arange(n)
        fresh = n
        rows_seen = []
        for f in range(6):
            world = synth.jitter_boxes(rng, world, 1.5)
            if f:
                gone = rng.choice(n, 120, replace=False)
                active[gone] = np.arange(fresh, fresh + 120)
                fresh += 120
            boxes = boxes_to_u2d(world[active])
            if kind == "visual":
                feats = synth.observe(rng, ident[active], 0.01)
                items = [TR.VisualSortObservation(feats[k], 0.9, boxes[k], None) for k in range(n)]
            else:
                items = [(boxes[k], None) for k in range(n)]
            rg, ro = g.predict(items), o.predict(items)
            assert_tracks_equal(rg, ro)
            cnt = C.c_uint32()
            g.lib.sa_tracks_count(g.lib.sa_tracker_engine(g.h), 0, C.byref(cnt))
            rows_seen.append(int(cnt.value))
        assert g.active_tracks() == o.active_tracks()
        assert max(rows_seen) > n and max(rows_seen) <= 2048, rows_seen
    finally:
        g.close()
        o.close()


@pytest.mark.gpu
@UPKEEP
@pytest.mark.parametrize("kind", ["sort", "visual"])
def test_frame_sizes_from_empty_to_hundreds_match_oracle(kind, backend):
    """predict() keeps its work arrays between calls: frames that shrink, grow and vanish (0, 9, 0, 260, 3, ...) must leave nothing of the
    previous frame behind — every frame's tracks against the oracle tracker's."""
    rng = np.random.default_rng(77)
    d, pool = 64, 260
    if kind == "visual":
        opts = (TR.VisualSortOptions().max_idle_epochs(2).kept_history_length(2).visual_metric(TR.VisualSortMetricType.cosine(0.5))
                .positional_metric(IoU(0.3)).visual_minimal_track_length(1).visual_max_observations(2).visual_min_votes(1))
        g, o = make(backend, "visual", opts=opts, feature_len=d), make("oracle", "visual", opts=opts, feature_len=d)
    else:
        kw = dict(bbox_history=2, max_idle_epochs=2, method=IoU(0.3), min_confidence=0.05)
        g, o = make(backend, "sort", **kw), make("oracle", "sort", **kw)
    try:
        ident = synth.reid_identities(rng, pool, d)
        world = synth.dense_boxes(rng, pool, (1400.0, 1000.0))
        for size in (0, 9, 0, 260, 3, 260, 0, 40, 41, 1):
            world = synth.jitter_boxes(rng, world, 2.0)
            pick = rng.permutation(pool)[:size]
            boxes = boxes_to_u2d(world[pick])
            if kind == "visual":
                feats = synth.observe(rng, ident[pick], 0.01) if size else np.zeros((0, d), np.float32)
                items = [TR.VisualSortObservation(ft, 0.9, bx, int(k) if k % 4 == 0 else None) for k, (bx, ft) in enumerate(zip(boxes, feats))]
            else:
                items = [(bx, int(k) if k % 4 == 0 else None) for k, bx in enumerate(boxes)]
            rg, ro = g.predict(items), o.predict(items)
            assert len(rg) == size
            assert_tracks_equal(rg, ro)
        assert g.active_tracks() == o.active_tracks()
    finally:
        g.close()
        o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sort", "visual"])
def test_queries_between_frames_see_the_finished_bookkeeping(kind):
    """With device upkeep the facade defers the heavy half of a frame's merges (history deques, observation policy) to the next call on
    the tracker.  Whatever that next call is — another predict(), idle_tracks, wasted, track_info, skip_epochs, active_tracks — it must
    see the finished state: a random interleaving of all of them, 120 frames of varying size, against the oracle tracker."""
    rng = np.random.default_rng(123)
    d, pool = 32, 90
    if kind == "visual":
        opts = (TR.VisualSortOptions().max_idle_epochs(2).kept_history_length(3).visual_metric(TR.VisualSortMetricType.cosine(0.5))
                .positional_metric(IoU(0.3)).visual_minimal_track_length(2).visual_max_observations(3).visual_min_votes(1)
                .visual_minimal_quality_use(0.3).visual_minimal_quality_collect(0.5))
        g, o = make("gpu_dev", "visual", opts=opts, feature_len=d), make("oracle", "visual", opts=opts, feature_len=d)
    else:
        kw = dict(bbox_history=3, max_idle_epochs=2, method=IoU(0.3), min_confidence=0.05)
        g, o = make("gpu_dev", "sort", **kw), make("oracle", "sort", **kw)
    try:
        ident = synth.reid_identities(rng, pool, d)
        world = synth.dense_boxes(rng, pool, (900.0, 700.0))
        seen = []
        for f in range(120):
            world = synth.jitter_boxes(rng, world, 2.0)
            size = int(rng.integers(0, pool + 1)) if f % 7 else 0
            pick = rng.permutation(pool)[:size]
            boxes = boxes_to_u2d(world[pick])
            if kind == "visual":
                feats = synth.observe(rng, ident[pick], 0.01) if size else np.zeros((0, d), np.float32)
                items = [TR.VisualSortObservation(None if k % 9 == 4 else ft, float(rng.uniform(0.2, 1.0)), bx, int(k) if k % 3 == 0 else None)
                         for k, (bx, ft) in enumerate(zip(boxes, feats))]
            else:
                items = [(bx, int(k) if k % 3 == 0 else None) for k, bx in enumerate(boxes)]
            rg, ro = g.predict(items), o.predict(items)
            assert_tracks_equal(rg, ro)
            seen = [x.id for x in rg] or seen
            # one or two queries, chosen at random, right behind the frame
            for _ in range(int(rng.integers(0, 3))):
                what = int(rng.integers(0, 6))
                if what == 0:
                    assert_tracks_equal(sorted(g.idle_tracks_with_scene(0), key=lambda x: x.id), sorted(o.idle_tracks_with_scene(0), key=lambda x: x.id))
                elif what == 1:
                    assert sorted(x.id for x in g.wasted()) == sorted(x.id for x in o.wasted())
                elif what == 2 and seen:
                    tid = seen[int(rng.integers(0, len(seen)))]   # (may have been wasted meanwhile: then both sides refuse)
                    info = []
                    for trk in (g, o):
                        try:
                            info.append(trk.track_info(tid))
                        except (TR.TrackerError, AssertionError):
                            info.append(None)
                    assert info[0] == info[1]
                elif what == 3:
                    n = int(rng.integers(1, 3))
                    g.skip_epochs_for_scene(0, n)
                    o.skip_epochs_for_scene(0, n)
                elif what == 4:
                    assert g.active_tracks() == o.active_tracks()
                else:
                    assert g.current_epoch_with_scene(0) == o.current_epoch_with_scene(0)
        assert sorted(x.id for x in g.wasted()) == sorted(x.id for x in o.wasted())
        assert g.active_tracks() == o.active_tracks()
    finally:
        g.close()
        o.close()


@pytest.mark.gpu
def test_caller_may_overwrite_its_device_feature_block_once_predict_has_returned():
    """The reference's predict() takes the observations' features by value.  Here a registered device block is READ IN PLACE, and the
    VisualSORT upkeep's bank dispatches are still running when predict() returns — they must not read the caller's block any more
    (the Kalman dispatch takes the rows into engine memory first).  Runs tests/devblock_overwrite_child.py (the device buffers are torch
    tensors, and torch's HIP context wants to be the first one of its process): two device-upkeep trackers see the same frames, one
    from a fresh region of device memory per frame, the other from ONE buffer that is overwritten — garbage, then the next frame —
    right after every predict() has returned; their tracks and their feature banks must be the same."""
    import os
    import subprocess
    import sys

    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "devblock_overwrite_child.py")
    r = subprocess.run([sys.executable, child], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OVERWRITE-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---- Batch*::predict over several scenes: one set of launches, the scenes' host work side by side ----------------------------------
def run_batch_visual_scenes(backend, seed, scenes=(4, 9, 17), sizes=(40, 70, 25), frames=8, d=64, bank=3, async_handle=False, reference="oracle",
                            **tracker_kw):
    """BatchVisualSort over several scenes of different sizes (visual_sort/batch_api.rs:213-317): every frame's tracks of every scene
    against the oracle tracker; with async_handle the request goes through sa_tracker_predict_batch_begin and the scenes are taken from
    the PredictionBatchResult handle in whatever order they finish."""
    rng = np.random.default_rng(seed)
    opts = (TR.VisualSortOptions().max_idle_epochs(2).kept_history_length(3).visual_metric(TR.VisualSortMetricType.cosine(0.5))
            .positional_metric(IoU(0.3)).visual_minimal_track_length(min(2, bank)).visual_minimal_area(500.0)
            .visual_minimal_quality_use(0.4).visual_minimal_quality_collect(0.6).visual_max_observations(bank).visual_min_votes(1))
    g = make(backend, "visual", opts=opts, feature_len=d, batch=True, **tracker_kw)
    o = make(reference, "visual", opts=opts, feature_len=d, batch=True)   # (the oracle tracker, or the facade's other upkeep path)
    try:
        ident = {s: synth.reid_identities(rng, n, d) for s, n in zip(scenes, sizes)}
        world = {s: synth.dense_boxes(rng, n, (900.0, 700.0)) for s, n in zip(scenes, sizes)}
        saw_visual = False
        for f in range(frames):
            req = TR.PredictionBatchRequest()
            for s, n in zip(scenes, sizes):
                if f == 3 and s == scenes[1]:
                    continue                        # a scene sits a frame out: its epoch stays, its tracks idle
                world[s] = synth.jitter_boxes(rng, world[s], 2.0)
                keep = rng.permutation(n)[rng.uniform(size=n) > 0.1]
                feats = synth.observe(rng, ident[s][keep], 0.01)
                for k, (bx, ft) in enumerate(zip(boxes_to_u2d(world[s][keep]), feats)):
                    q = float(rng.uniform(0.2, 1.0))
                    req.add(s, TR.VisualSortObservation(None if k % 7 == 3 else ft, None if k % 5 == 0 else q, bx, k if k % 2 else None))
            ro = o.predict_batch(req)
            if async_handle:
                h = g.predict_batch_async(req)
                assert h.batch_size() == len(req.scenes)
                rg = {}
                for _ in range(h.batch_size()):
                    sid, tracks = h.get()
                    assert sid not in rg
                    rg[sid] = tracks
                with pytest.raises(TR.TrackerError):
                    h.get()                         # every scene has been taken
                h.close()
            else:
                rg = g.predict_batch(req)
            assert sorted(rg) == sorted(ro)
            for s in rg:
                assert_tracks_equal(rg[s], ro[s])
                saw_visual = saw_visual or any(x.voting_type == TR.VotingType.Visual for x in rg[s])
                for x in rg[s][:5]:
                    assert g.track_info(x.id) == o.track_info(x.id)
            if f % 3 == 2:
                for s in scenes:
                    assert_tracks_equal(sorted(g.idle_tracks_with_scene(s), key=lambda x: x.id),
                                        sorted(o.idle_tracks_with_scene(s), key=lambda x: x.id))
        assert saw_visual
        assert g.active_tracks() == o.active_tracks()
    finally:
        g.close()
        o.close()


@pytest.mark.gpu
@UPKEEP
def test_batch_visual_sort_three_scenes_match_oracle(backend):
    run_batch_visual_scenes(backend, seed=61)


@pytest.mark.gpu
def test_long_batch_visual_run_device_upkeep_against_the_oracle_tracker():
    """150 frames of BatchVisualSort over three scenes through the result handle: the fused device-upkeep path (Kalman and feature banks
    on the GPU — row-major bank and its fragment-order twin —, completion words, the collect's table side ahead of the Kalman wait)
    against the ORACLE tracker, every frame — tracks, vote types, observation counts, idle sets."""
    run_batch_visual_scenes("gpu_dev", seed=89, frames=150, async_handle=True, reference="oracle")


@pytest.mark.gpu
def test_batch_visual_run_device_upkeep_against_host_upkeep():
    """... and 40 frames of the same against the facade's own host-upkeep path (two implementations of the same upkeep side by side)."""
    run_batch_visual_scenes("gpu_dev", seed=90, frames=40, async_handle=True, reference="gpu")


@pytest.mark.gpu
def test_batch_visual_sort_single_observation_scenes_match_oracle():
    """bank depth 1: the fused first phase votes itself, three scenes in one launch."""
    run_batch_visual_scenes("gpu_dev", seed=63, bank=1, sizes=(70, 33, 90))


@pytest.mark.gpu
@UPKEEP
def test_batch_result_handle_delivers_every_scene(backend):
    """sa_tracker_predict_batch_begin -> PredictionBatchResult (trackers/batch.rs:19-38): batch_size(), get() once per scene in completion
    order, the same tracks as the synchronous call's (both against the oracle tracker)."""
    run_batch_visual_scenes(backend, seed=67, async_handle=True, frames=6)


@pytest.mark.gpu
def test_long_batch_run_device_upkeep_against_the_oracle_tracker_and_host_upkeep():
    """400 frames of BatchSort over five scenes with missed detections, false positives and departures — the fused path (association with
    the upkeep queued behind it, completion words, the table side of the collect ahead of the Kalman wait, staged evictions, every third
    frame through the result handle) against the ORACLE tracker AND the facade's host-upkeep path, every frame: same tracks, same idle and
    wasted sets."""
    rng = np.random.default_rng(83)
    kw = dict(bbox_history=3, max_idle_epochs=2, method=IoU(0.3), min_confidence=0.05, batch=True)
    g, h, o = make("gpu_dev", "sort", **kw), make("gpu", "sort", **kw), make("oracle", "sort", **kw)
    try:
        scenes = (2, 3, 5, 7, 11)
        world = {s: synth.dense_boxes(rng, 70 + 10 * k, (1100.0, 800.0)) for k, s in enumerate(scenes)}
        for f in range(400):
            req = TR.PredictionBatchRequest()
            for s in scenes:
                world[s] = synth.jitter_boxes(rng, world[s], 2.0)
                keep = rng.uniform(size=len(world[s])) > 0.1
                extra = synth.dense_boxes(rng, int(rng.integers(0, 5)), (1100.0, 800.0))
                det = np.concatenate([world[s][keep], extra])
                det = det[rng.permutation(len(det))]
                for i, bx in enumerate(boxes_to_u2d(det)):
                    req.add(s, (bx, i if i % 4 == 0 else None))
                if f % 50 == 49:                                  # a few objects leave, as many arrive
                    world[s] = np.concatenate([world[s][3:], synth.dense_boxes(rng, 3, (1100.0, 800.0))])
            if f % 3 == 2:
                res = g.predict_batch_async(req)
                rg = {}
                for _ in range(res.batch_size()):
                    sid, tracks = res.get()
                    rg[sid] = tracks
                res.close()
            else:
                rg = g.predict_batch(req)
            rh = h.predict_batch(req)
            ro = o.predict_batch(req)
            for s in scenes:
                assert_tracks_equal(rg[s], ro[s])
                assert_tracks_equal(rg[s], rh[s])
            if f % 40 == 39:
                wo = sorted(x.id for x in o.wasted())
                assert sorted(x.id for x in g.wasted()) == wo and sorted(x.id for x in h.wasted()) == wo
                for s in scenes:
                    io = sorted(o.idle_tracks_with_scene(s), key=lambda x: x.id)
                    assert_tracks_equal(sorted(g.idle_tracks_with_scene(s), key=lambda x: x.id), io)
                    assert_tracks_equal(sorted(h.idle_tracks_with_scene(s), key=lambda x: x.id), io)
        assert g.active_tracks() == h.active_tracks() == o.active_tracks()
    finally:
        g.close()
        h.close()
        o.close()


# ---- one tracker object over several devices (sa_tracker_options.n_devices / devices): sort/batch_api.rs:157-207 -------------------
GROUPS = [pytest.param([0, 0], id="two_shards_one_gpu"), pytest.param([0, 0, 0, 0], id="four_shards_one_gpu")]


@pytest.mark.gpu
@pytest.mark.parametrize("devices", GROUPS)
@pytest.mark.parametrize("async_handle", [False, True], ids=["sync", "handle"])
def test_device_group_batch_visual_sort_matches_oracle(devices, async_handle):
    """BatchVisualSort with device upkeep on a GROUP of engines (here: several engines on the one GPU of the box; scenes dealt out scene_id
    % n, every shard's share through its own fused launches) against the oracle tracker: same ids — the id counter is the group's —, same
    tracks, vote types, idle sets, track_info through the group."""
    run_batch_visual_scenes("gpu_dev", seed=71, scenes=(4, 9, 17, 22, 31), sizes=(40, 70, 25, 33, 50), frames=10, devices=devices,
                            async_handle=async_handle)


@pytest.mark.gpu
@pytest.mark.parametrize("devices", GROUPS)
def test_device_group_churned_batch_sort_matches_oracle(devices):
    """Five churned BatchSort scenes (evictions, wasted tracks) over a group of engines, against the oracle tracker."""
    run_churned_batch_sort(workers=2, devices=devices)


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gpu", "gpu_dev"], ids=["host_upkeep", "device_upkeep"])
def test_device_group_with_sort_id_rules_hands_the_counter_from_scene_to_scene(backend):
    """Sort (NOT Batch*) id rules on a group of three engines: an id per track that STARTS, so the counter a scene starts from depends on
    the scenes before it — single-scene predicts and a multi-scene request both reproduce the oracle's ids; skip_epochs, wasted and
    active_tracks reach every shard."""
    rng = np.random.default_rng(5)
    kw = dict(bbox_history=3, max_idle_epochs=1, method=IoU(0.3), min_confidence=0.05)
    g, o = make(backend, "sort", devices=[0, 0, 0], **kw), make("oracle", "sort", **kw)
    try:
        scenes = (1, 2, 3, 7)
        world = {s: synth.dense_boxes(rng, 30 + 5 * s, (900.0, 700.0)) for s in scenes}
        for f in range(8):
            for s in scenes:
                world[s] = synth.jitter_boxes(rng, world[s], 2.0)
                keep = rng.uniform(size=len(world[s])) > 0.15
                det = [(bx, None) for bx in boxes_to_u2d(world[s][keep])]
                assert_tracks_equal(g.predict_with_scene(s, det), o.predict_with_scene(s, det))
            if f == 4:
                g.skip_epochs_for_scene(2, 3)
                o.skip_epochs_for_scene(2, 3)
                assert g.current_epoch_with_scene(2) == o.current_epoch_with_scene(2)
                assert sorted(x.id for x in g.wasted()) == sorted(x.id for x in o.wasted())
        req = TR.PredictionBatchRequest()
        for s in scenes:
            for bx in boxes_to_u2d(synth.jitter_boxes(rng, world[s], 2.0)):
                req.add(s, (bx, None))
        rg, ro = g.predict_batch(req), o.predict_batch(req)
        for s in scenes:
            assert_tracks_equal(rg[s], ro[s])
        assert g.active_tracks() == o.active_tracks()
    finally:
        g.close()
        o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("async_handle", [False, True], ids=["sync", "handle"])
def test_batch_trackers_with_no_idle_polling_match_the_oracle(async_handle):
    """Host manners: spin_us = 0 (no thread of the tracker polls while idle: the pool's workers, the result driver and a caller inside
    get() sleep at once) and workers = -4 (four threads left to the scheduler, no CPU claimed): the three-scene BatchVisualSort sequence
    and the churned five-scene BatchSort loop against the oracle tracker, as with the defaults."""
    run_batch_visual_scenes("gpu_dev", seed=73, frames=8, async_handle=async_handle, spin_us=0, workers=-4)
    run_churned_batch_sort(workers=-4, spin_us=0)


@pytest.mark.gpu
def test_device_group_refuses_a_bad_request_before_any_shard_has_begun():
    """A bad box in ONE shard's scene fails the whole call and no shard has moved: epochs, ids and track counts are those of a tracker
    that never saw the request; the next good request gives the oracle's tracks."""
    rng = np.random.default_rng(9)
    kw = dict(bbox_history=3, max_idle_epochs=2, method=IoU(0.3), min_confidence=0.05, batch=True)
    g, o = make("gpu_dev", "sort", devices=[0, 0], **kw), make("oracle", "sort", **kw)
    try:
        world = {s: synth.dense_boxes(rng, 20, (600.0, 400.0)) for s in (2, 3)}
        def request(bad=False):
            req = TR.PredictionBatchRequest()
            for s in (2, 3):
                for k, bx in enumerate(boxes_to_u2d(world[s])):
                    if bad and s == 3 and k == 7:
                        bx = TR.Universal2DBox(bx.xc, bx.yc, None, -1.0, bx.height, bx.confidence)   # aspect <= 0
                    req.add(s, (bx, None))
            return req
        rg, ro = g.predict_batch(request()), o.predict_batch(request())
        for s in (2, 3):
            assert_tracks_equal(rg[s], ro[s])
        with pytest.raises(TR.TrackerError):
            g.predict_batch(request(bad=True))
        assert g.active_tracks() == o.active_tracks() == 40
        assert g.current_epoch_with_scene(2) == o.current_epoch_with_scene(2) == 1
        for s in (2, 3):
            world[s] = synth.jitter_boxes(rng, world[s], 1.5)
        rg, ro = g.predict_batch(request()), o.predict_batch(request())
        for s in (2, 3):
            assert_tracks_equal(rg[s], ro[s])
    finally:
        g.close()
        o.close()


@pytest.mark.gpu
def test_batch_result_handle_survives_the_next_call_and_reused_request_arrays():
    """The request is taken by value: the caller's observation arrays are overwritten right after _begin returns, the next predict() is
    issued before the handle has been read (it waits for the set in flight — the reference's busy monitor), and the handle still
    delivers the first request's tracks."""
    rng = np.random.default_rng(71)
    kw = dict(bbox_history=3, max_idle_epochs=2, method=IoU(0.3), min_confidence=0.05, batch=True)
    g, o = make("gpu_dev", "sort", **kw), make("oracle", "sort", **kw)
    try:
        scenes, n = (1, 2, 3, 5, 8), 50
        world = {s: synth.dense_boxes(rng, n, (900.0, 700.0)) for s in scenes}
        lib = g.lib
        for f in range(5):
            req = TR.PredictionBatchRequest()
            for s in scenes:
                world[s] = synth.jitter_boxes(rng, world[s], 2.0)
                for k, bx in enumerate(boxes_to_u2d(world[s])):
                    req.add(s, (bx, k if k % 3 == 0 else None))
            ro = o.predict_batch(req)
            keep = []
            arrs = [g._obs_array(req.scenes[s], keep) for s in scenes]
            ids = (C.c_uint64 * len(scenes))(*scenes)
            counts = (C.c_uint32 * len(scenes))(*([n] * len(scenes)))
            pa = (C.POINTER(abi.sa_observation) * len(scenes))(*[C.cast(a, C.POINTER(abi.sa_observation)) for a in arrs])
            h = C.c_void_p()
            assert lib.sa_tracker_predict_batch_begin(g.h, len(scenes), ids, counts, pa, C.byref(h)) == 0
            for a in arrs:                                  # the caller's arrays are its own again
                C.memset(a, 0xFF, C.sizeof(a))
            assert g.current_epoch_with_scene(scenes[0]) == f + 1   # (any other entry point waits for the set in flight)
            assert lib.sa_batch_result_size(h) == len(scenes)
            got = {}
            out = (abi.sa_sort_track * n)()
            sid, cnt = C.c_uint64(), C.c_uint32()
            small = (abi.sa_sort_track * 1)()
            assert lib.sa_batch_result_get(h, C.byref(sid), small, 1, C.byref(cnt)) == abi.SA_ERR_BAD_ARG and cnt.value == n   # too small: nothing taken
            for _ in scenes:
                assert lib.sa_batch_result_ready(h) == 1    # (the set has finished: the call above waited for it)
                assert lib.sa_batch_result_get(h, C.byref(sid), out, n, C.byref(cnt)) == 0
                got[sid.value] = [TR.SortTrack.from_c(out[i]) for i in range(cnt.value)]
            assert lib.sa_batch_result_ready(h) == 0
            lib.sa_batch_result_free(h)
            for s in scenes:
                assert_tracks_equal(got[s], ro[s])
    finally:
        g.close()
        o.close()


@pytest.mark.gpu
@UPKEEP
def test_batch_request_without_scenes_is_a_no_op(backend):
    kw = dict(bbox_history=3, max_idle_epochs=2, method=IoU(0.3), min_confidence=0.05, batch=True)
    g = make(backend, "sort", **kw)
    try:
        assert g.predict_batch(TR.PredictionBatchRequest()) == {}
        req = TR.PredictionBatchRequest()
        req.add(3, (TR.Universal2DBox(10.0, 10.0, None, 1.0, 5.0), None))
        r = g.predict_batch(req)
        assert len(r[3]) == 1 and r[3][0].length == 1
        assert g.predict_batch(TR.PredictionBatchRequest()) == {}
        h = g.predict_batch_async(TR.PredictionBatchRequest())
        assert h.batch_size() == 0 and not h.ready()
        h.close()
        assert g.active_tracks() == 1
    finally:
        g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("workers", [1, 4], ids=["one_thread", "four_threads"])
def test_batch_sort_scenes_with_churn_and_eviction_match_oracle(workers):
    """Five scenes of 220 objects, ~18 % of them replaced every frame: the facade evicts expired rows from SEVERAL scenes' tables inside one
    predict() (the removals are queued one behind the other), with the scenes' bookkeeping spread over a pool of threads — every frame's
    tracks of every scene against the oracle tracker."""
    run_churned_batch_sort(workers=workers)


def run_churned_batch_sort(**tracker_kw):
    rng = np.random.default_rng(909)
    scenes, n = (2, 3, 5, 7, 11), 220
    o_, keep_ = TR.sort_options(3, 2, IoU(0.3), 0.05, None, 1.0 / 20.0, 1.0 / 160.0, batch=True, device_upkeep=True, **tracker_kw)
    g = TR._Tracker(o_, keep_)
    group = len(tracker_kw.get("devices") or []) > 1
    o = make("oracle", "sort", bbox_history=3, max_idle_epochs=2, method=IoU(0.3), min_confidence=0.05, batch=True)
    try:
        pool = n + 14 * 40
        world = {s: synth.dense_boxes(rng, pool, (2600.0, 1800.0)) for s in scenes}
        active = {s: np.arange(n) for s in scenes}
        fresh = {s: n for s in scenes}
        max_rows = 0
        for f in range(14):
            req = TR.PredictionBatchRequest()
            for s in scenes:
                world[s] = synth.jitter_boxes(rng, world[s], 1.5)
                if f:
                    gone = rng.choice(n, 40, replace=False)
                    active[s][gone] = np.arange(fresh[s], fresh[s] + 40)
                    fresh[s] += 40
                for bx in boxes_to_u2d(world[s][active[s]]):
                    req.add(s, (bx, None))
            rg, ro = g.predict_batch(req), o.predict_batch(req)
            for s in scenes:
                assert_tracks_equal(rg[s], ro[s])
                cnt = C.c_uint32()
                if group:   # (sa_tracker_engine hands out a group's FIRST engine: the scene may live on another)
                    cnt.value = n + 1
                else:
                    g.lib.sa_tracks_count(g.lib.sa_tracker_engine(g.h), s, C.byref(cnt))
                max_rows = max(max_rows, int(cnt.value))
                assert cnt.value < n + 6 * 40
            if f % 5 == 4:
                for s in scenes:
                    assert_tracks_equal(sorted(g.idle_tracks_with_scene(s), key=lambda x: x.id), sorted(o.idle_tracks_with_scene(s), key=lambda x: x.id))
        assert max_rows > n
        assert g.active_tracks() == o.active_tracks()
        assert sorted(x.id for x in g.wasted()) == sorted(x.id for x in o.wasted())
    finally:
        g.close()
        o.close()
