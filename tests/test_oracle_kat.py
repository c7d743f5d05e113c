"""Pins the CPU oracle against every known-answer test the reference holds for the hot path
(SURVEY.md §4 / Appendix C).  The expected values are the literals asserted by the reference's own
#[test] functions; file:line cited per test.  CPU only."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_lib as O
from similari_amd import abi

EPS = 1e-5
L = O.lib()


def U(xc, yc, angle, aspect, height, conf=1.0):
    return abi.make_boxes([xc], [yc], [aspect], [height], confidence=[conf], angle=None if angle is None else [angle])


def feat(v):
    v = np.asarray(v, np.float32)
    out = np.zeros(8 * L.or_feature_blocks(len(v)), np.float32)
    nb = L.or_feature_pad(O.fptr(v) if len(v) else None, len(v), O.fptr(out))
    return out, nb


# ---- src/distance.rs:57-78 ----------------------------------------------------------------------
def test_distance_kat():
    v1, b1 = feat([1, 0, 0])
    v2, b2 = feat([0, 1, 0])
    v3, b3 = feat([-1, 0, 0])
    assert abs(L.or_euclidean(O.fptr(v1), b1, O.fptr(v1), b1)) < EPS
    assert abs(L.or_euclidean(O.fptr(v1), b1, O.fptr(v2), b2) - math.sqrt(2)) < EPS
    assert abs(L.or_cosine(O.fptr(v1), b1, O.fptr(v1), b1) - 1.0) < EPS
    assert abs(L.or_cosine(O.fptr(v1), b1, O.fptr(v3), b3) + 1.0) < EPS
    assert abs(L.or_cosine(O.fptr(v1), b1, O.fptr(v2), b2)) < EPS


# ---- src/track/utils.rs:85-90 -------------------------------------------------------------------
def test_feature_padding_kat():
    out, nb = feat([0.0, 0.2, 0.3])
    assert nb == 1
    np.testing.assert_array_equal(out, np.array([0.0, 0.2, 0.3, 0, 0, 0, 0, 0], np.float32))
    out, nb = feat([])
    assert nb == 1 and not out.any()
    _, nb = feat(np.arange(16))
    assert nb == 2
    _, nb = feat(np.arange(17))
    assert nb == 3


def test_distance_length_mismatch_truncates():
    # distance.rs:11,28: min(len1, len2) blocks
    a, ba = feat(np.arange(1, 17))
    b, bb = feat(np.arange(1, 9))
    assert abs(L.or_euclidean(O.fptr(a), ba, O.fptr(b), bb)) < EPS
    assert abs(L.or_cosine(O.fptr(a), ba, O.fptr(b), bb) - 1.0) < EPS


def test_cosine_zero_vector_is_nan():
    a, ba = feat([0, 0, 0])
    b, bb = feat([1, 0, 0])
    assert math.isnan(L.or_cosine(O.fptr(a), ba, O.fptr(b), bb))


# ---- src/utils/bbox.rs:823-870 ------------------------------------------------------------------
def test_radius_too_far_dist_in_2r_kat():
    b1 = abi.ltwh(0, 0, 6, 8)
    assert abs(L.or_radius(O.box_ptr(b1)) - 5.0) < EPS
    b2 = abi.ltwh(6, 0, 6, 8)
    assert not L.or_too_far(O.box_ptr(b1), O.box_ptr(b2))
    assert abs(L.or_dist_in_2r(O.box_ptr(b1), O.box_ptr(b2)) - 0.6) < EPS
    b3 = abi.ltwh(10, 0, 6, 8)
    assert not L.or_too_far(O.box_ptr(b1), O.box_ptr(b3))
    assert abs(L.or_dist_in_2r(O.box_ptr(b1), O.box_ptr(b3)) - 1.0) < EPS
    b4 = abi.ltwh(10.1, 0, 6, 8)
    assert L.or_too_far(O.box_ptr(b1), O.box_ptr(b4))
    assert not L.or_too_far(O.box_ptr(b1), O.box_ptr(b1))
    assert abs(L.or_dist_in_2r(O.box_ptr(b1), O.box_ptr(b1))) < EPS


# ---- src/utils/bbox.rs:873-884 (BoundingBox IoU: >0.999, >0.8, <0.001) via the oriented path ---
def test_axis_aligned_iou_kat():
    b1 = abi.ltwh(-1.0, -1.0, 2.0, 2.0)
    out = C.c_float()
    assert L.or_iou(O.box_ptr(b1), O.box_ptr(b1), C.byref(out)) == 1
    assert out.value > 0.999
    b2 = abi.ltwh(-0.9, -0.9, 2.0, 2.0)
    assert L.or_iou(O.box_ptr(b1), O.box_ptr(b2), C.byref(out)) == 1
    assert out.value > 0.8
    assert L.or_iou(O.box_ptr(b2), O.box_ptr(b2), C.byref(out)) == 1 and out.value > 0.999
    b3 = abi.ltwh(1.0, 1.0, 3.0, 3.0)
    for a in (b1, b2):
        r = L.or_iou(O.box_ptr(a), O.box_ptr(b3), C.byref(out))
        assert r == 0 or out.value < 0.001


# ---- src/utils/bbox.rs:341-369 ------------------------------------------------------------------
def test_oriented_clip_and_iou_kat():
    pi = np.float32(np.pi)
    a2 = np.float32(2.0) + pi / np.float32(2.0)
    b1 = U(0.0, 0.0, 2.0, 0.5, 2.0)
    b2 = U(0.0, 0.0, a2, 0.5, 2.0)
    p1 = np.zeros(8)
    p2 = np.zeros(8)
    L.or_vertices(O.box_ptr(b1), O.dptr(p1))
    L.or_vertices(O.box_ptr(b2), O.dptr(p2))
    out = np.zeros(32)
    n = L.or_sh_clip(O.dptr(p1), 4, O.dptr(p2), 4, O.dptr(out))
    int_area = L.or_polygon_area(O.dptr(out), n)
    assert abs(int_area - 1.0) < EPS  # 1x2 and 2x1 rectangles crossing at the centre
    union = 2.0 + 2.0 - int_area
    assert abs(union - 3.0) < EPS
    iou = C.c_float()
    assert L.or_iou(O.box_ptr(b1), O.box_ptr(b2), C.byref(iou)) == 1
    assert abs(iou.value - int_area / union) < EPS
    assert abs(iou.value - 1.0 / 3.0) < EPS
    b3 = U(10.0, 0.0, a2, 0.5, 2.0)
    assert L.or_iou(O.box_ptr(b1), O.box_ptr(b3), C.byref(iou)) == 0
    assert L.or_intersection(O.box_ptr(b1), O.box_ptr(b3)) == 0.0


# ---- src/utils/clipping.rs:98-115, bbox.rs:371-380 (no panic / finite) --------------------------
def test_clip_corner_cases_run():
    subj = np.array([8055.658, 7977.5537, 8010.734, 7999.9697, 8032.9717, 8044.537, 8077.896, 8022.121])
    clip = np.array([8055.805, 7977.847, 8010.871, 8000.2676, 8033.105, 8044.8286, 8078.039, 8022.408])
    out = np.zeros(32)
    n = L.or_sh_clip(O.dptr(subj), 4, O.dptr(clip), 4, O.dptr(out))
    assert 3 <= n <= 8
    a = L.or_polygon_area(O.dptr(out), n)
    assert np.isfinite(a) and a > 0
    x = U(8044.315, 8011.0454, 2.6787748, 1.00801, 49.8073)
    y = U(8044.455, 8011.338, 2.6787748, 1.0083783, 49.79979)
    iou = C.c_float()
    assert L.or_iou(O.box_ptr(x), O.box_ptr(y), C.byref(iou)) == 1
    assert 0.9 < iou.value <= 1.0


def test_polygon_area_properties():
    sq = np.array([0, 0, 0, 2, 2, 2, 2, 0], float)
    assert L.or_polygon_area(O.dptr(sq), 4) == 4.0
    assert L.or_polygon_area(O.dptr(sq[::-1].copy()), 4) == 4.0
    assert L.or_polygon_area(O.dptr(sq), 2) == 0.0
    assert L.or_polygon_area(O.dptr(sq), 0) == 0.0
    closed = np.array([0, 0, 0, 2, 2, 2, 2, 0, 0, 0], float)
    assert L.or_polygon_area(O.dptr(closed), 5) == 4.0


# ---- src/utils/kalman/kalman_2d_box.rs:193-249 --------------------------------------------------
PW, VW = np.float32(1.0 / 20.0), np.float32(1.0 / 160.0)


def kf_init(box):
    m = np.zeros(10, np.float32)
    c = np.zeros(100, np.float32)
    L.or_kf_initiate(PW, VW, O.box_ptr(box), O.fptr(m), O.fptr(c))
    return m, c


def kf_predict(m, c):
    m2 = np.zeros(10, np.float32)
    c2 = np.zeros(100, np.float32)
    L.or_kf_predict(PW, VW, O.fptr(m), O.fptr(c), O.fptr(m2), O.fptr(c2))
    return m2, c2


def kf_update(m, c, z):
    m2 = np.zeros(10, np.float32)
    c2 = np.zeros(100, np.float32)
    L.or_kf_update(PW, VW, O.fptr(m), O.fptr(c), O.box_ptr(z), O.fptr(m2), O.fptr(c2))
    return m2, c2


def box_eq(m, exp):
    # Universal2DBox PartialEq  bbox.rs:537-545
    return (
        abs(m[0] - exp[0]) < EPS
        and abs(m[1] - exp[1]) < EPS
        and (m[2] - exp[2]) < EPS
        and (m[3] - exp[3]) < EPS
        and (m[4] - exp[4]) < EPS
    )


def test_kalman_constructor_kat():
    b = abi.ltwh(1.0, 2.0, 5.0, 5.0)
    m, _ = kf_init(b)
    assert box_eq(m, [3.5, 4.5, 0.0, 1.0, 5.0])


def test_kalman_step_kat():
    m, c = kf_init(abi.ltwh(-10.0, 2.0, 2.0, 5.0))
    m, c = kf_predict(m, c)
    assert box_eq(m, [-9.0, 4.5, 0.0, 0.4, 5.0])
    m, c = kf_update(m, c, U(8.75, 52.35, None, 0.15084915, 100.1))
    m, c = kf_predict(m, c)
    exp = [10.070248, 55.90909, 0.0, 0.3951147, 107.173546]
    assert box_eq(m, exp), m[:5]
    # the scratch numpy restatement in SURVEY reproduced these bit-for-bit in f32:
    assert np.float32(m[0]) == np.float32(10.070248)
    assert np.float32(m[1]) == np.float32(55.90909)
    assert np.float32(m[3]) == np.float32(0.3951147)
    assert np.float32(m[4]) == np.float32(107.173546)


def test_kalman_gating_kat():
    m, c = kf_init(abi.ltwh(-10.0, 2.0, 2.0, 5.0))
    m, c = kf_predict(m, c)
    m, c = kf_update(m, c, abi.ltwh(-9.5, 2.1, 2.0, 5.0))
    m, c = kf_predict(m, c)
    d1 = L.or_kf_distance(PW, VW, O.fptr(m), O.fptr(c), O.box_ptr(abi.ltwh(-9.0, 2.2, 2.0, 5.0)))
    d1c = L.or_kf_cost(d1, 0)
    assert 0.0 <= d1c < 11.070
    assert abs(d1 - 0.7857) < 1e-3
    d2 = L.or_kf_distance(PW, VW, O.fptr(m), O.fptr(c), O.box_ptr(abi.ltwh(-5.0, 1.5, 2.2, 5.0)))
    assert L.or_kf_cost(d2, 0) > 11.070
    assert abs(d2 - 74.89) < 0.05
    # the 5/25 projection the C ABI carries gives the same number
    m5 = m[:5].copy()
    c25 = c.reshape(10, 10)[:5, :5].copy().ravel()
    d1b = L.or_kf_distance5(PW, O.fptr(m5), O.fptr(c25), O.box_ptr(abi.ltwh(-9.0, 2.2, 2.0, 5.0)))
    assert d1b == d1


def test_kalman_cost():
    assert L.or_kf_cost(5.0, 1) == 95.0
    assert L.or_kf_cost(12.0, 1) == 0.0
    assert L.or_kf_cost(12.0, 0) == 100.0
    assert L.or_kf_cost(5.0, 0) == 5.0


# ---- src/trackers/spatio_temporal_constraints.rs (tests :100-121) -------------------------------
def test_constraints_validate():
    d = np.array([1, 2], np.uint64)
    m = np.array([0.5, 1.0], np.float32)
    dp = d.ctypes.data_as(C.POINTER(C.c_uint64))
    assert L.or_constraints_validate(2, dp, O.fptr(m), 1, 0.4)
    assert not L.or_constraints_validate(2, dp, O.fptr(m), 1, 0.6)
    assert L.or_constraints_validate(2, dp, O.fptr(m), 2, 0.9)
    assert not L.or_constraints_validate(2, dp, O.fptr(m), 2, 1.1)
    assert L.or_constraints_validate(2, dp, O.fptr(m), 3, 100.0)  # no constraint with delta >= 3
    assert L.or_constraints_validate(0, None, None, 1, 100.0)


def test_constraints_kat():
    # spatio_temporal_constraints.rs:100-121: two add_constraints calls, stable sort + dedup keeps the first
    cfg = abi.make_config(constraints=[(1, 0.5), (2, 1.0), (3, 2.0), (4, 4.0), (3, 2.5), (4, 4.5), (7, 8.5)])
    assert cfg.n_constraints == 5
    v = lambda d, x: L.or_constraints_validate(cfg.n_constraints, cfg.constraint_epoch_delta, cfg.constraint_max_dist, d, x)
    assert v(1, 0.4) and not v(1, 0.6)
    assert v(3, 2.0) and not v(3, 2.4)
    assert v(6, 7.0) and not v(6, 9.0)
    assert v(7, 8.4) and v(7, 8.5) and not v(7, 8.7)
    assert v(9, 8.7) and v(9, 100.0)


# ---- src/trackers/sort/metric.rs:162-218 --------------------------------------------------------
def test_sort_metric_confidence_kat():
    cfg = abi.make_config(positional="iou", positional_threshold=0.3, positional_min_confidence=0.05)
    cand = abi.ltwh(0.0, 0.0, 8.0, 10.0, confidence=0.8)
    track = abi.ltwh(0.0, 0.0, 8.0, 10.0, confidence=1.0)
    out = C.c_float()
    assert L.or_positional_metric(C.byref(cfg), O.box_ptr(cand), O.box_ptr(track), None, None, C.byref(out)) == 1
    assert abs(out.value - 0.8) < EPS
    assert L.or_positional_metric(C.byref(cfg), O.box_ptr(track), O.box_ptr(cand), None, None, C.byref(out)) == 1
    assert abs(out.value - 1.0) < EPS


def test_quantise():
    assert L.or_quantise(0.3) == 300000
    assert L.or_quantise(0.6) == 600000
    assert L.or_quantise(float("nan")) == 0
    assert L.or_quantise(1e30) == 2**63 - 1
    assert L.or_quantise(-1e30) == -(2**63)
    assert L.or_quantise(-0.5) == -500000


# ---- src/trackers/sort/voting.rs:110-174 --------------------------------------------------------
def _u64(a):
    a = np.asarray(a, np.uint64)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint64))


def test_sort_voting_kat():
    rows = [
        (10, 20, 0.6), (10, 25, 0.4), (10, 30, 0.4),
        (11, 20, 0.5), (11, 25, 0.69), (11, 30, 0.4),
        (12, 20, 0.2), (12, 25, 0.27), (12, 30, 0.28),
    ]
    fr, frp = _u64([r[0] for r in rows])
    to, top = _u64([r[1] for r in rows])
    w = np.array([r[2] for r in rows], np.float32)
    ids, idp = _u64([10, 11, 12])
    out, outp = _u64([0, 0, 0])
    total = C.c_int64()
    L.or_sort_voting(0.3, 3, 3, len(rows), frp, top, O.fptr(w), 3, idp, outp, C.byref(total))
    assert list(out) == [20, 25, 12]
    assert total.value == 1_590_000


def test_kuhn_munkres_matches_scipy_total():
    from scipy.optimize import linear_sum_assignment

    rng = np.random.default_rng(0)
    for _ in range(50):
        r = int(rng.integers(1, 9))
        c = int(rng.integers(r, 13))
        w = rng.integers(-50, 1000, size=(r, c)).astype(np.int64)
        total = C.c_int64()
        assign = np.zeros(r, np.uint32)
        rc = L.or_kuhn_munkres(r, c, w.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(total), assign.ctypes.data_as(C.POINTER(C.c_uint32)))
        assert rc == 0
        ri, ci = linear_sum_assignment(w, maximize=True)
        assert total.value == w[ri, ci].sum()
        assert len(set(assign.tolist())) == r
        assert w[np.arange(r), assign].sum() == total.value


# ---- src/trackers/visual_sort/voting.rs:109-202 -------------------------------------------------
NAN = float("nan")


def _vv(thr, maxd, votes, rows, ids):
    fr, frp = _u64([r[0] for r in rows])
    to, top = _u64([r[1] for r in rows])
    pos = np.array([r[2] for r in rows], np.float32)
    vis = np.array([r[3] for r in rows], np.float32)
    idarr, idp = _u64(ids)
    out, outp = _u64([0] * len(ids))
    typ = np.zeros(len(ids), np.uint8)
    L.or_visual_voting(thr, maxd, votes, len(rows), frp, top, O.fptr(pos), O.fptr(vis), len(ids), idp, outp, typ.ctypes.data_as(C.POINTER(C.c_uint8)))
    return {int(i): (int(o), int(t)) for i, o, t in zip(ids, out, typ) if o}


VIS, POS = abi.SA_VOTE_VISUAL, abi.SA_VOTE_POSITIONAL


def test_visual_voting_kat():
    assert _vv(0.3, 0.7, 1, [(1, 2, 0.7, 0.7)], [1]) == {1: (2, VIS)}
    assert _vv(0.3, 0.7, 2, [(1, 2, 0.7, 0.7)], [1]) == {1: (2, POS)}
    five = [(1, 2, 0.7, 0.7), (1, 2, NAN, 0.68), (1, 2, NAN, 0.65), (1, 3, 0.7, 0.7), (1, 3, NAN, 0.64)]
    assert _vv(0.3, 0.7, 2, five, [1]) == {1: (2, VIS)}
    comp = five[:3] + [(4, 3, 0.7, 0.7), (4, 3, NAN, 0.64)]
    assert _vv(0.3, 0.7, 2, comp, [1, 4]) == {1: (2, VIS), 4: (3, VIS)}
    assert _vv(0.3, 0.7, 2, five + [(11, 2, 0.8, 0.7), (11, 3, 0.6, 0.64)], [1, 11]) == {1: (2, VIS), 11: (3, POS)}
    assert _vv(0.3, 0.7, 2, five + [(11, 2, 0.8, 0.7), (11, 3, NAN, 0.64)], [1, 11]) == {1: (2, VIS)}


# ---- src/track/voting/topn.rs:142-281 — same Σ(max−d) weight formula as BestFit (best.rs:93) ----
def _bf(rows, ids, votes=1, maxd=3.4e38):
    fr, frp = _u64([r[0] for r in rows])
    to, top = _u64([r[1] for r in rows])
    vis = np.array([r[2] for r in rows], np.float32)
    idarr, idp = _u64(ids)
    out, outp = _u64([0] * len(ids))
    w = np.zeros(len(ids))
    L.or_bestfit_voting(maxd, votes, len(rows), frp, top, O.fptr(vis), len(ids), idp, outp, O.dptr(w))
    return list(out), list(w)


def test_bestfit_weight_formula_kat():
    out, w = _bf([(7, 1, 0.2)], [7])
    assert out == [1] and w == [0.0]
    out, w = _bf([(7, 1, 0.2), (7, 1, 0.3)], [7])
    assert out == [1] and w[0] == 0.10000000894069672
    out, w = _bf([(7, 1, 0.2), (7, 1, 0.4)], [7], maxd=0.32)
    assert out == [1] and w[0] == 0.20000000298023224


def test_bestfit_greedy_marks_every_group():
    # best.rs:106-120: every group of a candidate marks its track taken, not only its best
    rows = [(1, 10, 0.1), (1, 20, 0.2), (2, 20, 0.3), (2, 30, 0.4), (3, 99, 0.9)]
    out, _ = _bf(rows, [1, 2, 3])
    assert out == [10, 2, 99]  # candidate 2's best (track 20) was taken by candidate 1's second group -> self


# ---- src/trackers/visual_sort/metric.rs:675-1098 (VisualMetric gating matrix) -------------------
def _vm(cfg, cand_box, cand_feat, track_box, track_feats, present=None, quality=1.0, own_area=None, track_kf=None):
    K = cfg.max_observations
    D = cfg.feature_len
    tf = np.zeros((1, K, D), np.float32)
    pr = np.zeros((1, K), np.uint8)
    for k, f in enumerate(track_feats):
        tf[0, k] = f
        pr[0, k] = 1
    if present is not None:
        pr[0, :] = present
    if track_kf is None:
        m, c = kf_init(track_box)
        track_kf = (m[:5].copy(), c.reshape(10, 10)[:5, :5].copy().ravel())
    tracks = abi.make_tracks([5], track_box, [0], kf_mean=track_kf[0], kf_cov=track_kf[1], feats=tf, feat_present=pr)
    det = abi.make_detections(cand_box, feats=np.asarray([cand_feat], np.float32), feat_quality=[quality],
                              own_area=None if own_area is None else [own_area])
    r = O.associate(cfg, tracks, 0, det)
    return r["positional"][0, 0], r["visual"][0, 0]


def test_visual_metric_gating_kat():
    some = lambda v: v == v
    # Maha, far boxes, euclid(MAX): (None, Some>0)   metric.rs:675-724
    cfg = abi.make_config(positional="maha", visual="euclidean", visual_threshold=3.4e38, feature_len=2,
                          max_observations=3, visual_minimal_track_length=1, positional_min_confidence=0.1)
    p, v = _vm(cfg, abi.ltwh(100.3, 0.3, 5.1, 10.0), [0.1, 1.1], abi.ltwh(0.3, 0.3, 5.1, 10.0), [[0.1, 1.0]])
    assert not some(p) and some(v[0]) and v[0] > 0
    # IoU(.3)+cosine(1.0), identical boxes, feats (1,0): (~1.0, ~0.0)   metric.rs:726-773
    cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual="cosine", visual_threshold=1.0,
                          feature_len=2, max_observations=3, visual_minimal_track_length=1, positional_min_confidence=0.1)
    b = abi.ltwh(0.3, 0.3, 5.1, 10.0)
    p, v = _vm(cfg, b, [1.0, 0.0], b, [[1.0, 0.0]])
    assert abs(p - 1.0) < EPS and abs(v[0]) < EPS
    # Maha+euclid(10), identical: (~100.0, ~0.0)   metric.rs:775-823
    cfg = abi.make_config(positional="maha", visual="euclidean", visual_threshold=10.0, feature_len=2,
                          max_observations=3, visual_minimal_track_length=1, positional_min_confidence=0.1)
    p, v = _vm(cfg, b, [1.0, 0.0], b, [[1.0, 0.0]])
    assert abs(p - 100.0) < 1e-3 and abs(v[0]) < EPS
    # IoU, min_len 3, one observation each: (~1.0, None)   metric.rs:825-880
    cfg = abi.make_config(positional="iou", visual="euclidean", visual_threshold=3.4e38, feature_len=2,
                          max_observations=3, visual_minimal_track_length=3, positional_min_confidence=0.1)
    p, v = _vm(cfg, b, [1.0, 0.0], b, [[1.0, 0.0]])
    assert abs(p - 1.0) < EPS and not some(v[0])
    # min_len 2, track has two identical obs: rows [(~1.0, ~0), (None, ~0)]
    cfg = abi.make_config(positional="iou", visual="euclidean", visual_threshold=3.4e38, feature_len=2,
                          max_observations=3, visual_minimal_track_length=2, positional_min_confidence=0.1)
    p, v = _vm(cfg, b, [1.0, 0.0], b, [[1.0, 0.0], [1.0, 0.0]])
    assert abs(p - 1.0) < EPS and abs(v[0]) < EPS and abs(v[1]) < EPS and not some(v[2])
    # small candidate box with visual_minimal_area(1.0): (None, None)
    cfg = abi.make_config(positional="iou", visual="euclidean", visual_threshold=3.4e38, feature_len=2,
                          max_observations=3, visual_minimal_track_length=1, visual_minimal_area=1.0,
                          positional_min_confidence=0.1)
    p, v = _vm(cfg, abi.ltwh(0.3, 0.3, 0.8, 1.0), [1.0, 0.0], b, [[1.0, 0.0]])
    assert not some(p) and not some(v[0])
    # candidate quality .2 with visual_minimal_quality_use(.3): (~1.0, None)
    cfg = abi.make_config(positional="iou", visual="euclidean", visual_threshold=3.4e38, feature_len=2,
                          max_observations=3, visual_minimal_track_length=1, visual_minimal_quality_use=0.3,
                          positional_min_confidence=0.1)
    p, v = _vm(cfg, b, [1.0, 0.0], b, [[1.0, 0.0]], quality=0.2)
    assert abs(p - 1.0) < EPS and not some(v[0])
    # candidate own-area .5 with own_area_percentage_use(.6): (~1.0, None)
    cfg = abi.make_config(positional="iou", visual="euclidean", visual_threshold=3.4e38, feature_len=2,
                          max_observations=3, visual_minimal_track_length=1,
                          visual_minimal_own_area_percentage_use=0.6, positional_min_confidence=0.1)
    p, v = _vm(cfg, b, [1.0, 0.0], b, [[1.0, 0.0]], own_area=0.5)
    assert abs(p - 1.0) < EPS and not some(v[0])
    p, v = _vm(cfg, b, [1.0, 0.0], b, [[1.0, 0.0]], own_area=0.7)
    assert abs(p - 1.0) < EPS and some(v[0])


# ---- frame-level sanity: identity association --------------------------------------------------
def test_associate_identity_sort_iou():
    rng = np.random.default_rng(1)
    n = 40
    xs = np.arange(n) * 100.0
    boxes = abi.make_boxes(xs, xs * 0 + 50, np.full(n, 0.5), np.full(n, 80.0), confidence=rng.uniform(0.5, 1, n))
    cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
    tracks = abi.make_tracks(np.arange(1, n + 1), boxes, np.zeros(n))
    perm = rng.permutation(n)
    det = abi.make_detections(boxes[perm])
    r = O.associate(cfg, tracks, 1, det)
    np.testing.assert_array_equal(r["track_id"], perm + 1)
    assert (r["voting_type"] == POS).all()
    # idle tracks are masked
    r = O.associate(cfg, tracks, 7, det)
    assert (r["track_id"] == 0).all()


# ---- the sharded oracle (or_associate_sharded): distance stage on host threads partitioned like store.rs:490-493, one vote after
# the shards — must be the single-thread oracle, cell for cell and id for id ---------------------------------------------------
@pytest.mark.parametrize("visual", [None, "cosine", "euclidean"])
def test_sharded_oracle_equals_the_single_thread_oracle(visual):
    from similari_amd import synth

    rng = np.random.default_rng(77)
    n, t, d, k = 90, 110, 48, 2
    if visual:
        sc = synth.visual_scene(rng, t, n, d, k, canvas=(900.0, 700.0), new_fraction=0.2)
        sc["track_present"][rng.uniform(size=(t, k)) < 0.2] = 0
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual=visual, visual_threshold=0.4 if visual == "cosine" else 0.6,
                              feature_len=d, max_observations=k, visual_min_votes=1, visual_minimal_track_length=1, max_idle_epochs=5)
        tr = abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"], feats=sc["track_feats"], feat_present=sc["track_present"])
        det = abi.make_detections(sc["det_boxes"], feats=sc["det_feats"], feat_quality=sc["det_quality"])
    else:
        sc = synth.sort_scene(rng, t, n, canvas=(700.0, 500.0))
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
        tr = abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"])
        det = abi.make_detections(sc["det_boxes"])
    one = O.associate(cfg, tr, 1, det)
    for shards in (2, 7):
        many = O.associate(cfg, tr, 1, det, shards=shards)
        for key, v in one.items():
            if isinstance(v, np.ndarray):
                np.testing.assert_array_equal(v, many[key], err_msg=f"{key}, {shards} shards")
            else:
                assert v == many[key], (key, shards)
    assert (one["track_id"] != 0).sum() > 30
