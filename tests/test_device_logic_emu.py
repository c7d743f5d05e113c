"""CPU checks of the product's scalar device logic (similari_amd/csrc/sa_device.h compiled with g++ by
tests/emu) against the oracle: per-cell IoU / Mahalanobis / pre-filter arithmetic must be bit-identical, the
sparse per-component assignment must reach the dense kuhn_munkres optimum.  No GPU needed."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as O
from similari_amd import abi

ROOT = Path(__file__).resolve().parent.parent
EMU_DIR = ROOT / "tests" / "emu"


def emu():
    so = EMU_DIR / "libemu.so"
    srcs = [EMU_DIR / "emu.cpp", ROOT / "similari_amd" / "csrc" / "sa_device.h", ROOT / "similari_amd" / "csrc" / "sa_dense.h"]
    if not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        subprocess.run(
            ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-o", str(so), str(EMU_DIR / "emu.cpp")],
            check=True,
        )
    L = C.CDLL(str(so))
    B = C.POINTER(abi.sa_box)
    fp = C.POINTER(C.c_float)
    L.emu_positional_cell.restype = C.c_int
    L.emu_positional_cell.argtypes = [C.POINTER(abi.sa_config), B, C.c_uint64, B, C.c_uint64, fp, fp, fp, C.POINTER(C.c_int)]
    L.emu_quantise.restype = C.c_int64
    L.emu_quantise.argtypes = [C.c_float]
    L.emu_f32_key.restype = C.c_uint32
    L.emu_f32_key.argtypes = [C.c_float]
    L.emu_key_f32.restype = C.c_float
    L.emu_key_f32.argtypes = [C.c_uint32]
    L.emu_assign.restype = C.c_int
    L.emu_assign.argtypes = [C.c_uint32, C.c_uint32, fp, C.c_int64, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    L.emu_assign_coop.restype = C.c_int
    L.emu_assign_coop.argtypes = [C.c_uint32, C.c_uint32, fp, C.c_int64, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int, C.c_int,
                                  C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    L.emu_clip_is_empty.restype = C.c_int
    L.emu_clip_is_empty.argtypes = [B, B]
    L.emu_assign_dense.restype = C.c_int
    L.emu_assign_dense.argtypes = [C.c_uint32, C.c_uint32, fp, C.c_int64, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int, C.c_uint32,
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    return L


E = emu()
OL = O.lib()


def random_boxes(rng, n, canvas=400.0, oriented=False):
    b = abi.make_boxes(
        rng.uniform(0, canvas, n), rng.uniform(0, canvas, n), rng.uniform(0.3, 1.5, n), rng.uniform(20, 120, n),
        confidence=rng.uniform(0.0, 1.0, n), angle=rng.uniform(0, 3.2, n) if oriented else None,
    )
    return b


@pytest.mark.parametrize("oriented", [False, True])
def test_iou_cells_bit_identical(oriented):
    rng = np.random.default_rng(3 + oriented)
    n = 400
    a = random_boxes(rng, n, oriented=oriented)
    b = random_boxes(rng, n, oriented=oriented)
    # half of the pairs: near-duplicates (true matches), which stress the clipper's degenerate branches
    b[: n // 2] = a[: n // 2]
    b["xc"][: n // 2] += rng.normal(0, 2, n // 2).astype(np.float32)
    b["yc"][: n // 2] += rng.normal(0, 2, n // 2).astype(np.float32)
    b[:5] = a[:5]  # exact duplicates
    cfg = abi.make_config(positional="iou", positional_threshold=0.3, positional_min_confidence=0.05, max_idle_epochs=3,
                          constraints=[(1, 1.0), (3, 2.5)])
    present = 0
    for i in range(n):
        for te in (5, 3, 1):
            ov, ev = C.c_float(), C.c_float()
            comp = C.c_int()
            oc = OL.or_compatible(C.byref(cfg), O.box_ptr(a, i), 5, O.box_ptr(b, i), te)
            r_o = OL.or_positional_metric(C.byref(cfg), O.box_ptr(a, i), O.box_ptr(b, i), None, None, C.byref(ov)) if oc else 0
            r_e = E.emu_positional_cell(C.byref(cfg), O.box_ptr(a, i), 5, O.box_ptr(b, i), te, None, None, C.byref(ev), C.byref(comp))
            assert bool(comp.value) == bool(oc)
            assert r_o == r_e, (i, te)
            if r_o:
                present += 1
                assert np.float32(ov.value).tobytes() == np.float32(ev.value).tobytes()
                assert OL.or_quantise(ov.value) == E.emu_quantise(ev.value)
    assert present > 100


@pytest.mark.parametrize("thr", [0.05, 0.3, 0.7])
def test_axis_aligned_quick_reject_is_conservative(thr):
    """The positional tiles decide most axis-aligned pairs without the f64 clip (sa_aa_quick_reject: the rectangles miss each other, or
    an upper bound of IoU x confidence stays below the threshold).  Whatever it rejects must be absent in the oracle too: 60 000
    pairs aimed at the decision boundaries — rectangles that just touch, overlaps of a hair, IoU x confidence a hair either side of
    the threshold, nested boxes, huge coordinates."""
    rng = np.random.default_rng(int(thr * 100))
    cfg = abi.make_config(positional="iou", positional_threshold=thr, positional_min_confidence=0.05, max_idle_epochs=3)
    n = 20000
    present = rejected_nearly = 0
    for variant in range(3):
        a = random_boxes(rng, n, canvas=400.0 if variant < 2 else 30000.0)
        b = a.copy()
        w, h = a["aspect"] * a["height"], a["height"]
        if variant == 0:   # sliding along x until the rectangles (almost) stop touching
            b["aspect"] = rng.uniform(0.3, 1.5, n).astype(np.float32)
            b["height"] = rng.uniform(20, 120, n).astype(np.float32)
            gap = rng.choice([-1e-3, -1e-5, 0.0, 1e-5, 1e-3, 0.5], n).astype(np.float32)
            b["xc"] = a["xc"] + (w + b["aspect"] * b["height"]) / 2 + gap
            b["yc"] = a["yc"] + rng.uniform(-10, 10, n).astype(np.float32)
        else:              # a shift chosen so that IoU x confidence lands close to the threshold
            conf = a["confidence"].clip(0.05, 1.0)
            target = np.clip(thr / conf * rng.choice([0.98, 0.999, 0.99999, 1.0, 1.00001, 1.001, 1.02], n), 0.0, 0.999)
            # same-size boxes shifted by d along x: IoU = (w - d) / (w + d)  ->  d = w (1 - t) / (1 + t)
            b["xc"] = a["xc"] + (w * (1 - target) / (1 + target)).astype(np.float32)
        b["confidence"] = rng.uniform(0, 1, n).astype(np.float32)
        for i in range(n):
            ov, ev = C.c_float(), C.c_float()
            comp = C.c_int()
            r_o = OL.or_positional_metric(C.byref(cfg), O.box_ptr(a, i), O.box_ptr(b, i), None, None, C.byref(ov))
            r_e = E.emu_positional_cell(C.byref(cfg), O.box_ptr(a, i), 5, O.box_ptr(b, i), 5, None, None, C.byref(ev), C.byref(comp))
            assert r_o == r_e, (variant, i, a[i], b[i])
            if r_o:
                present += 1
                assert np.float32(ov.value).tobytes() == np.float32(ev.value).tobytes()
    assert present > 2000


def test_maha_cells_bit_identical():
    rng = np.random.default_rng(5)
    n = 300
    pw, vw = np.float32(1 / 20), np.float32(1 / 160)
    cfg = abi.make_config(positional="maha", positional_min_confidence=0.1)
    tracks = random_boxes(rng, n)
    hits = 0
    for i in range(n):
        m = np.zeros(10, np.float32)
        c = np.zeros(100, np.float32)
        OL.or_kf_initiate(pw, vw, O.box_ptr(tracks, i), O.fptr(m), O.fptr(c))
        for _ in range(int(rng.integers(1, 4))):  # a few predict/update steps -> realistic state
            m2, c2 = np.zeros(10, np.float32), np.zeros(100, np.float32)
            OL.or_kf_predict(pw, vw, O.fptr(m), O.fptr(c), O.fptr(m2), O.fptr(c2))
            z = tracks[i : i + 1].copy()
            z["xc"] += np.float32(rng.normal(0, 2))
            z["yc"] += np.float32(rng.normal(0, 2))
            OL.or_kf_update(pw, vw, O.fptr(m2), O.fptr(c2), O.box_ptr(z), O.fptr(m), O.fptr(c))
        tb = np.zeros(1, abi.BOX_DTYPE)
        OL.or_kf_state_box(O.fptr(m), O.box_ptr(tb))
        tb["confidence"] = 1.0
        cand = tb.copy()
        cand["xc"] += np.float32(rng.normal(0, 3))
        cand["yc"] += np.float32(rng.normal(0, 3))
        cand["confidence"] = np.float32(rng.uniform(0, 1))
        m5 = m[:5].copy()
        c25 = c.reshape(10, 10)[:5, :5].copy().ravel()
        ov, ev = C.c_float(), C.c_float()
        r_o = OL.or_positional_metric(C.byref(cfg), O.box_ptr(cand), O.box_ptr(tb), O.fptr(m5), O.fptr(c25), C.byref(ov))
        r_e = E.emu_positional_cell(C.byref(cfg), O.box_ptr(cand), 0, O.box_ptr(tb), 0, O.fptr(m5), O.fptr(c25), C.byref(ev), None)
        assert r_o == r_e
        if r_o:
            hits += 1
            assert np.float32(ov.value).tobytes() == np.float32(ev.value).tobytes(), (ov.value, ev.value)
    assert hits > 200


def test_maha_cells_off_diagonal_covariances_bit_identical():
    """The same cell on covariances with real off-diagonal mass (the filter's own stay diagonal: the inner loops of the Cholesky and of
    the forward substitution multiply zeros in the test above), and on ones whose pivot fails: NaN on both sides."""
    rng = np.random.default_rng(55)
    cfg = abi.make_config(positional="maha", positional_min_confidence=0.1)
    n = 400
    tracks = random_boxes(rng, n)
    hits = nans = 0
    for i in range(n):
        tb = tracks[i : i + 1].copy()
        tb["confidence"] = 1.0
        h = float(tb["height"][0])
        m5 = np.array([tb["xc"][0], tb["yc"][0], 0.0, tb["aspect"][0], h], np.float32)
        g = rng.standard_normal((5, 7))
        r = g @ g.T
        d = np.sqrt(np.diag(r))
        r = np.eye(5) + rng.uniform(0.2, 0.95) * (r / d[:, None] / d[None, :] - np.eye(5))
        sd = np.sqrt(np.array([(h / 20) ** 2 * 3, (h / 20) ** 2 * 3, 1e-2, 1e-2, (h / 20) ** 2 * 3]) * rng.uniform(0.3, 3.0, 5))
        c = (sd[:, None] * r * sd[None, :]).astype(np.float32)
        c = np.triu(c) + np.triu(c, 1).T
        if i % 25 == 7:
            c[int(rng.integers(0, 5))] *= np.float32(-50.0)   # not positive definite any more (and not symmetric): some pivot fails
        c25 = np.ascontiguousarray(c.ravel())
        cand = tb.copy()
        cand["xc"] += np.float32(rng.normal(0, 3))
        cand["yc"] += np.float32(rng.normal(0, 3))
        cand["confidence"] = np.float32(rng.uniform(0, 1))
        ov, ev = C.c_float(), C.c_float()
        r_o = OL.or_positional_metric(C.byref(cfg), O.box_ptr(cand), O.box_ptr(tb), O.fptr(m5), O.fptr(c25), C.byref(ov))
        r_e = E.emu_positional_cell(C.byref(cfg), O.box_ptr(cand), 0, O.box_ptr(tb), 0, O.fptr(m5), O.fptr(c25), C.byref(ev), None)
        assert r_o == r_e
        if r_o:
            hits += 1
            if np.isnan(ov.value):
                nans += 1
                assert np.isnan(ev.value)
            else:
                assert np.float32(ov.value).tobytes() == np.float32(ev.value).tobytes(), (i, ov.value, ev.value)
    assert hits > 300 and nans >= 3


def test_float_key_roundtrip_and_order():
    vals = np.array([-3.5, -1.0, -0.0, 0.0, 1e-30, 0.5, 1.0, 2.0, 3.4e38], np.float32)
    keys = [E.emu_f32_key(float(v)) for v in vals]
    assert keys == sorted(keys)
    for v, k in zip(vals, keys):
        assert np.float32(E.emu_key_f32(k)).tobytes() == np.float32(v).tobytes()
    assert all(k > 0 for k in keys)


def dense_reference(pos, thr_q):
    """SortVoting's dense N x (N+T) matrix through the oracle's kuhn_munkres."""
    N, T = pos.shape
    w = np.zeros((N, N + T), np.int64)
    for i in range(N):
        for j in range(T):
            if pos[i, j] == pos[i, j]:
                w[i, N + j] = OL.or_quantise(float(pos[i, j]))
        w[i, i] = thr_q
    total = C.c_int64()
    assign = np.zeros(N, np.uint32)
    OL.or_kuhn_munkres(N, N + T, w.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(total), assign.ctypes.data_as(C.POINTER(C.c_uint32)))
    ref = np.where(assign >= N, assign.astype(np.int64) - N, -1)
    return total.value, ref, w


def run_emu_assign(pos, thr_q, row_skip=None, col_skip=None):
    N, T = pos.shape
    pos = np.ascontiguousarray(pos, np.float32)
    rm = np.zeros(N, np.int32)
    tot = C.c_int64()
    rs = None if row_skip is None else row_skip.ctypes.data_as(C.POINTER(C.c_uint8))
    cs = None if col_skip is None else col_skip.ctypes.data_as(C.POINTER(C.c_uint8))
    rc = E.emu_assign(N, T, O.fptr(pos), thr_q, rs, cs, rm.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(tot))
    assert rc == 0, f"optimality certificate failed: {rc}"
    return rm, tot.value


@pytest.mark.parametrize("density", [0.02, 0.1, 0.5, 1.0])
def test_sparse_assignment_reaches_dense_optimum(density):
    rng = np.random.default_rng(int(density * 100))
    thr_q = 300000
    for trial in range(30):
        N = int(rng.integers(1, 40))
        T = int(rng.integers(1, 40))
        pos = rng.uniform(0.05, 1.0, (N, T)).astype(np.float32)
        pos[rng.uniform(size=(N, T)) > density] = np.nan
        rm, gain = run_emu_assign(pos, thr_q)
        total, ref, w = dense_reference(pos, thr_q)
        assert gain + N * thr_q == total
        # a matched edge always beats the threshold; unmatched rows are the "self" rows
        for i in range(N):
            if rm[i] >= 0:
                assert w[i, N + rm[i]] > thr_q
        # unique optimum (random f32 weights: ties have probability ~0) -> identical indices
        np.testing.assert_array_equal(rm, ref)


def test_sparse_assignment_kat_from_reference():
    # sort/voting.rs:110-174 : {10->20, 11->25, 12->self}
    pos = np.array([[0.6, 0.4, 0.4], [0.5, 0.69, 0.4], [0.2, 0.27, 0.28]], np.float32)
    rm, gain = run_emu_assign(pos, 300000)
    assert list(rm) == [0, 1, -1]
    assert gain + 3 * 300000 == 1_590_000


def test_sparse_assignment_long_chains_and_exclusions():
    # a chain r0-c0-r1-c1-...: augmenting paths must run through the whole component
    n = 60
    pos = np.full((n, n), np.nan, np.float32)
    rng = np.random.default_rng(9)
    for i in range(n):
        pos[i, i] = 0.5 + 0.001 * i
        if i + 1 < n:
            pos[i + 1, i] = 0.9 - 0.002 * i
    rm, gain = run_emu_assign(pos, 300000)
    total, ref, _ = dense_reference(pos, 300000)
    assert gain + n * 300000 == total
    np.testing.assert_array_equal(rm, ref)
    # exclusions behave like absent rows / columns
    row_skip = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    col_skip = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    rm2, gain2 = run_emu_assign(pos, 300000, row_skip, col_skip)
    p2 = pos.copy()
    p2[row_skip.astype(bool), :] = np.nan
    p2[:, col_skip.astype(bool)] = np.nan
    total2, ref2, _ = dense_reference(p2, 300000)
    assert gain2 + n * 300000 == total2
    np.testing.assert_array_equal(rm2, ref2)


def test_sparse_assignment_ties_keep_total():
    # integer-valued weights -> many ties: totals must still agree (indices are unpinned on ties)
    rng = np.random.default_rng(11)
    for _ in range(40):
        N, T = int(rng.integers(2, 25)), int(rng.integers(2, 25))
        pos = (rng.integers(1, 6, (N, T)) / 5.0).astype(np.float32)
        pos[rng.uniform(size=(N, T)) > 0.4] = np.nan
        rm, gain = run_emu_assign(pos, 300000)
        total, _, _ = dense_reference(pos, 300000)
        assert gain + N * 300000 == total
        used = [c for c in rm if c >= 0]
        assert len(used) == len(set(used))


def test_mahalanobis_scale_weights():
    # Mahalanobis weights reach 100/min_conf * 1e6 ~ 1e9..1e10: i64 arithmetic end to end
    rng = np.random.default_rng(13)
    for _ in range(20):
        N, T = int(rng.integers(2, 20)), int(rng.integers(2, 20))
        pos = (rng.uniform(88.0, 100.0, (N, T)) / rng.uniform(0.05, 1.0, (N, 1))).astype(np.float32)
        pos[rng.uniform(size=(N, T)) > 0.3] = np.nan
        rm, gain = run_emu_assign(pos, 1000000)
        total, ref, _ = dense_reference(pos, 1000000)
        assert gain + N * 1000000 == total
        np.testing.assert_array_equal(rm, ref)


@pytest.mark.parametrize("oriented", [False, True])
def test_clip_prefilter_only_skips_empty_intersections(oriented):
    """sa_clip_is_empty (the positional tiles' pre-filter in the heterogeneous launch) may only fire where the reference's clip
    returns exactly 0.0 — and it should fire for a good share of the bounding-circle neighbours, or it is useless."""
    rng = np.random.default_rng(40 + oriented)
    n = 180
    a = random_boxes(rng, n, canvas=700.0, oriented=oriented)
    b = random_boxes(rng, n, canvas=700.0, oriented=oriented)
    # near-touching pairs: boxes placed edge to edge with gaps from 1e-9 to 1 px, the band where a wrong margin would show
    t = a[:100].copy()
    gap = 10.0 ** rng.uniform(-9, 0, 100)
    t["xc"] = (a[:100]["xc"] + (a[:100]["aspect"] * a[:100]["height"] + t["aspect"] * t["height"]) / 2 + gap * rng.choice([-1, 1], 100)).astype(np.float32)
    b[:100] = t
    near = skipped = wrong = 0
    B = C.POINTER(abi.sa_box)
    for i in range(n):
        for j in range(n):
            pa, pb = a[i : i + 1].ctypes.data_as(B), b[j : j + 1].ctypes.data_as(B)
            if OL.or_too_far(pa, pb):
                continue
            near += 1
            if E.emu_clip_is_empty(pa, pb):
                skipped += 1
                wrong += OL.or_intersection(pa, pb) != 0.0
    assert wrong == 0
    assert skipped > 0.3 * near, (skipped, near)


# ---- the group-cooperative solver (sa_assign_component_coop: greedy start + lane-parallel shortest augmenting paths) ----------
def run_emu_assign_coop(pos, thr_q, G, row_skip=None, col_skip=None, hbm_lists=0):
    N, T = pos.shape
    pos = np.ascontiguousarray(pos, np.float32)
    rm = np.zeros(max(N, 1), np.int32)
    tot = C.c_int64()
    rs = None if row_skip is None else row_skip.ctypes.data_as(C.POINTER(C.c_uint8))
    cs = None if col_skip is None else col_skip.ctypes.data_as(C.POINTER(C.c_uint8))
    rc = E.emu_assign_coop(N, T, O.fptr(pos), thr_q, rs, cs, G, hbm_lists, rm.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(tot))
    assert rc == 0, f"optimality certificate failed: {rc}"
    return rm[:N], tot.value


@pytest.mark.parametrize("G", [4, 16, 64])
@pytest.mark.parametrize("density", [0.02, 0.1, 0.5, 1.0])
def test_cooperative_assignment_reaches_dense_optimum(density, G):
    rng = np.random.default_rng(int(density * 100) + G)
    thr_q = 300000
    for trial in range(25):
        N = int(rng.integers(1, 70))
        T = int(rng.integers(1, 70))
        pos = rng.uniform(0.05, 1.0, (N, T)).astype(np.float32)
        pos[rng.uniform(size=(N, T)) > density] = np.nan
        rm, gain = run_emu_assign_coop(pos, thr_q, G)
        total, ref, w = dense_reference(pos, thr_q)
        assert gain + N * thr_q == total
        np.testing.assert_array_equal(rm, ref)  # unique optimum (random f32 weights)
        rm_serial, gain_serial = run_emu_assign(pos, thr_q)
        np.testing.assert_array_equal(rm, rm_serial)


@pytest.mark.parametrize("G", [16, 64])
def test_cooperative_assignment_one_giant_component(G):
    """All boxes on one pile under a low threshold: one component of 200 rows, every row with ~100 usable edges, most bids of the
    greedy start colliding — the case the one-thread-per-component solver cannot afford."""
    rng = np.random.default_rng(7 + G)
    n = t = 200
    pos = rng.uniform(0.06, 0.9, (n, t)).astype(np.float32)
    pos[rng.uniform(size=(n, t)) > 0.5] = np.nan
    rm, gain = run_emu_assign_coop(pos, 50000, G)
    total, ref, _ = dense_reference(pos, 50000)
    assert gain + n * 50000 == total
    np.testing.assert_array_equal(rm, ref)


@pytest.mark.parametrize("hbm_lists", [0, 1])
@pytest.mark.parametrize("G", [4, 64])
def test_cooperative_assignment_chains_exclusions_and_ties(G, hbm_lists):
    n = 60
    pos = np.full((n, n), np.nan, np.float32)
    rng = np.random.default_rng(9)
    for i in range(n):
        pos[i, i] = 0.5 + 0.001 * i
        if i + 1 < n:
            pos[i + 1, i] = 0.9 - 0.002 * i
    rm, gain = run_emu_assign_coop(pos, 300000, G, hbm_lists=hbm_lists)
    total, ref, _ = dense_reference(pos, 300000)
    assert gain + n * 300000 == total
    np.testing.assert_array_equal(rm, ref)
    row_skip = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    col_skip = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    rm2, gain2 = run_emu_assign_coop(pos, 300000, G, row_skip, col_skip, hbm_lists=hbm_lists)
    p2 = pos.copy()
    p2[row_skip.astype(bool), :] = np.nan
    p2[:, col_skip.astype(bool)] = np.nan
    total2, ref2, _ = dense_reference(p2, 300000)
    assert gain2 + n * 300000 == total2
    np.testing.assert_array_equal(rm2, ref2)
    # integer-valued weights -> many ties: the total still agrees, every column is used once (indices are unpinned on ties)
    for _ in range(30):
        N, T = int(rng.integers(2, 40)), int(rng.integers(2, 40))
        pt = (rng.integers(1, 6, (N, T)) / 5.0).astype(np.float32)
        pt[rng.uniform(size=(N, T)) > 0.4] = np.nan
        rm3, gain3 = run_emu_assign_coop(pt, 300000, G, hbm_lists=hbm_lists)
        total3, _, _ = dense_reference(pt, 300000)
        assert gain3 + N * 300000 == total3
        used = [c for c in rm3 if c >= 0]
        assert len(used) == len(set(used))
    # Mahalanobis-scale weights: i64 end to end
    for _ in range(10):
        N, T = int(rng.integers(2, 30)), int(rng.integers(2, 30))
        pm = (rng.uniform(88.0, 100.0, (N, T)) / rng.uniform(0.05, 1.0, (N, 1))).astype(np.float32)
        pm[rng.uniform(size=(N, T)) > 0.3] = np.nan
        rm4, gain4 = run_emu_assign_coop(pm, 1000000, G, hbm_lists=hbm_lists)
        total4, ref4, _ = dense_reference(pm, 1000000)
        assert gain4 + N * 1000000 == total4
        np.testing.assert_array_equal(rm4, ref4)


# ---- the workgroup-cooperative DENSE solver (sa_dense.h: one thread per column strip, dense row reads, packed-key minima) -------
def run_emu_assign_dense(pos, thr_q, shape=1, min_roots=1, row_skip=None, col_skip=None):
    N, T = pos.shape
    pos = np.ascontiguousarray(pos, np.float32)
    rm = np.zeros(max(N, 1), np.int32)
    tot = C.c_int64()
    rs = None if row_skip is None else row_skip.ctypes.data_as(C.POINTER(C.c_uint8))
    cs = None if col_skip is None else col_skip.ctypes.data_as(C.POINTER(C.c_uint8))
    rc = E.emu_assign_dense(N, T, O.fptr(pos), thr_q, rs, cs, shape, min_roots, rm.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(tot))
    assert rc == 0, f"dense solver: inconsistent matching ({rc})"
    return rm[:N], tot.value


@pytest.mark.parametrize("shape", [1, 2, 17, 18])
@pytest.mark.parametrize("density", [0.02, 0.1, 0.5, 1.0])
def test_dense_assignment_reaches_dense_optimum(density, shape):
    rng = np.random.default_rng(int(density * 100) + 1000 * shape)
    thr_q = 300000
    for trial in range(25):
        N = int(rng.integers(1, 90))
        T = int(rng.integers(1, 90))
        pos = rng.uniform(0.05, 1.0, (N, T)).astype(np.float32)
        pos[rng.uniform(size=(N, T)) > density] = np.nan
        rm, gain = run_emu_assign_dense(pos, thr_q, shape)
        total, ref, w = dense_reference(pos, thr_q)
        assert gain + N * thr_q == total
        np.testing.assert_array_equal(rm, ref)  # unique optimum (random f32 weights)
        rm_coop, gain_coop = run_emu_assign_coop(pos, thr_q, 64)
        np.testing.assert_array_equal(rm, rm_coop)
        # mixed, as on the device: small components to the wavefront solver, the others to the dense one
        rm_mix, gain_mix = run_emu_assign_dense(pos, thr_q, shape, min_roots=3)
        np.testing.assert_array_equal(rm_mix, ref)


@pytest.mark.parametrize("shape", [4, 5])
@pytest.mark.parametrize("density", [0.05, 0.3, 1.0])
def test_dense_assignment_one_wavefront_form(density, shape):
    """The general tail's middle tier: sa_assign_component_dense<64, 2 | 4> on a [rows][128 | 256] matrix of 32-bit cells (one
    wavefront, the component's columns renumbered).  Same matchings as the dense kuhn_munkres and as the 256-thread form; ties
    break alike."""
    rng = np.random.default_rng(int(density * 100) + 4000 + shape)
    thr_q = 300000
    for trial in range(30):
        N, T = int(rng.integers(1, 64 if shape == 4 else 33)), int(rng.integers(1, 129 if shape == 4 else 257))
        pos = rng.uniform(0.05, 1.0, (N, T)).astype(np.float32)
        pos[rng.uniform(size=(N, T)) > density] = np.nan
        rm, gain = run_emu_assign_dense(pos, thr_q, shape)
        total, ref, _ = dense_reference(pos, thr_q)
        assert gain + N * thr_q == total
        np.testing.assert_array_equal(rm, ref)
        pt = (rng.integers(1, 6, (N, T)) / 5.0).astype(np.float32)
        pt[rng.uniform(size=(N, T)) > max(density, 0.3)] = np.nan
        rm_t, gain_t = run_emu_assign_dense(pt, thr_q, shape)
        rm_w, gain_w = run_emu_assign_dense(pt, thr_q, 1)
        assert gain_t == gain_w
        np.testing.assert_array_equal(rm_t, rm_w)


@pytest.mark.parametrize("shape,n,t", [(1, 300, 320), (3, 640, 700), (19, 640, 700), (1, 1024, 1024)])
def test_dense_assignment_one_giant_component(shape, n, t):
    """All boxes on one pile under a low threshold: one component of hundreds of rows, half of all cells usable, most greedy bids
    colliding (the `giant` / `bigpile` bench frames).  Same matching as the dense kuhn_munkres and as the wavefront solver."""
    rng = np.random.default_rng(7 + n)
    pos = rng.uniform(0.06, 0.9, (n, t)).astype(np.float32)
    pos[rng.uniform(size=(n, t)) > 0.5] = np.nan
    rm, gain = run_emu_assign_dense(pos, 50000, shape)
    total, ref, _ = dense_reference(pos, 50000)
    assert gain + n * 50000 == total
    np.testing.assert_array_equal(rm, ref)


@pytest.mark.parametrize("shape", [1, 2, 18])
def test_dense_assignment_chains_exclusions_and_ties(shape):
    n = 60
    pos = np.full((n, n), np.nan, np.float32)
    rng = np.random.default_rng(9)
    for i in range(n):
        pos[i, i] = 0.5 + 0.001 * i
        if i + 1 < n:
            pos[i + 1, i] = 0.9 - 0.002 * i
    rm, gain = run_emu_assign_dense(pos, 300000, shape)
    total, ref, _ = dense_reference(pos, 300000)
    assert gain + n * 300000 == total
    np.testing.assert_array_equal(rm, ref)
    row_skip = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    col_skip = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    rm2, gain2 = run_emu_assign_dense(pos, 300000, shape, row_skip=row_skip, col_skip=col_skip)
    p2 = pos.copy()
    p2[row_skip.astype(bool), :] = np.nan
    p2[:, col_skip.astype(bool)] = np.nan
    total2, ref2, _ = dense_reference(p2, 300000)
    assert gain2 + n * 300000 == total2
    np.testing.assert_array_equal(rm2, ref2)
    # integer-valued weights -> many ties: the total still agrees, every column is used once (indices are unpinned on ties), and
    # the tie-breaks are the wavefront solver's (same (distance, column) order): identical matchings
    for _ in range(30):
        N, T = int(rng.integers(2, 40)), int(rng.integers(2, 40))
        pt = (rng.integers(1, 6, (N, T)) / 5.0).astype(np.float32)
        pt[rng.uniform(size=(N, T)) > 0.4] = np.nan
        rm3, gain3 = run_emu_assign_dense(pt, 300000, shape)
        total3, _, _ = dense_reference(pt, 300000)
        assert gain3 + N * 300000 == total3
        used = [c for c in rm3 if c >= 0]
        assert len(used) == len(set(used))
        rm3c, _ = run_emu_assign_coop(pt, 300000, 64)
        np.testing.assert_array_equal(rm3, rm3c)
    # Mahalanobis-scale weights: i64 end to end, keys beyond 32 bits
    for _ in range(10):
        N, T = int(rng.integers(2, 30)), int(rng.integers(2, 30))
        pm = (rng.uniform(88.0, 100.0, (N, T)) / rng.uniform(0.05, 1.0, (N, 1))).astype(np.float32)
        pm[rng.uniform(size=(N, T)) > 0.3] = np.nan
        rm4, gain4 = run_emu_assign_dense(pm, 1000000, shape)
        total4, ref4, _ = dense_reference(pm, 1000000)
        assert gain4 + N * 1000000 == total4
        np.testing.assert_array_equal(rm4, ref4)
