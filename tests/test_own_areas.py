"""Exclusively owned areas (src/utils/clipping/bbox_own_areas.rs:8-46).

cpu : the oracle's or_own_area_shares on the reference's own unit test (bbox_own_areas.rs:58-82), against a raster estimate of
      the same set difference (an independent third method), and on properties of the definition.
gpu : sa_own_areas through the C ABI against the oracle to the reference test's own tolerance (EPS = 1e-5): the device
      integrates the boundary of the owned region, the oracle decomposes it into convex pieces — different algorithms, so
      agreement is evidence for both."""
import numpy as np
import pytest

import oracle_lib as O
from similari_amd import abi, synth

EPS = 1e-5


def emu_shares(boxes, max_nb=127, cap=24):
    """The device's scalar logic (sa_device.h: sa_own_edge & co) compiled for the host by tests/emu."""
    import ctypes as C
    from test_device_logic_emu import E
    E.emu_own_areas.restype = C.c_int
    E.emu_own_areas.argtypes = [C.c_uint32, C.POINTER(abi.sa_box), C.POINTER(C.c_float), C.c_uint32, C.c_uint32]
    boxes = np.ascontiguousarray(boxes, abi.BOX_DTYPE)
    out = np.zeros(max(len(boxes), 1), np.float32)
    st = E.emu_own_areas(len(boxes), C.cast(boxes.ctypes.data, C.POINTER(abi.sa_box)), out.ctypes.data_as(C.POINTER(C.c_float)),
                         max_nb, cap)
    return out[: len(boxes)].copy(), st


def raster_shares(boxes, res=400):
    """Brute force: sample each box on a res x res grid of its own frame, count samples inside no other box."""
    n = len(boxes)
    out = np.zeros(n)
    u = (np.arange(res) + 0.5) / res - 0.5
    U, V = np.meshgrid(u, u)
    for i in range(n):
        b = boxes[i]
        w, h = float(b["aspect"]) * float(b["height"]), float(b["height"])
        a = float(b["angle"]) if b["has_angle"] else 0.0
        X = float(b["xc"]) + U * w * np.cos(a) - V * h * np.sin(a)
        Y = float(b["yc"]) + U * w * np.sin(a) + V * h * np.cos(a)
        free = np.ones_like(X, bool)
        for j in range(n):
            if j == i:
                continue
            c = boxes[j]
            wj, hj = float(c["aspect"]) * float(c["height"]), float(c["height"])
            aj = float(c["angle"]) if c["has_angle"] else 0.0
            dx, dy = X - float(c["xc"]), Y - float(c["yc"])
            lx = dx * np.cos(aj) + dy * np.sin(aj)
            ly = -dx * np.sin(aj) + dy * np.cos(aj)
            free &= ~((np.abs(lx) <= wj / 2) & (np.abs(ly) <= hj / 2))
        out[i] = free.mean()
    return out


def test_reference_unit_example():
    # bbox_own_areas.rs:60-81: ltwh (0,0,10,10), (5,5,10,10), (10,10,10,10) own 75, 50, 75 of 100
    b = np.concatenate([abi.ltwh([0.0], [0.0], [10.0], [10.0]), abi.ltwh([5.0], [5.0], [10.0], [10.0]),
                        abi.ltwh([10.0], [10.0], [10.0], [10.0])])
    s = O.own_area_shares(b)
    assert np.abs(s - np.array([0.75, 0.50, 0.75], np.float32)).max() < EPS


def test_definition_properties():
    rng = np.random.default_rng(1)
    far = synth.dense_boxes(rng, 30, (60000.0, 60000.0))
    s = O.own_area_shares(far)
    # nothing overlaps: area / (area + EPS) just below 1
    assert (s <= 1.0).all() and (s > 1.0 - 1e-4).all()
    # a box and its exact duplicate own nothing
    dup = np.concatenate([far[:5], far[:5]])
    assert np.abs(O.own_area_shares(dup)).max() < EPS
    # a small box inside a big one: the small one owns nothing, the big one everything but the small one
    big, small = abi.ltwh([0.0], [0.0], [100.0], [50.0]), abi.ltwh([10.0], [10.0], [20.0], [10.0])
    s = O.own_area_shares(np.concatenate([big, small]))
    assert abs(s[1]) < EPS and abs(s[0] - (5000.0 - 200.0) / 5000.0) < EPS
    # empty and single inputs
    assert len(O.own_area_shares(far[:0])) == 0
    assert O.own_area_shares(far[:1])[0] > 1.0 - 1e-4


@pytest.mark.parametrize("oriented", [False, True])
def test_oracle_against_raster(oriented):
    rng = np.random.default_rng(7 + oriented)
    b = synth.dense_boxes(rng, 60, (900.0, 600.0), oriented=oriented)
    s = O.own_area_shares(b)
    r = raster_shares(b)
    assert (s < 0.999).sum() > 20, "the scene must actually overlap"
    assert np.abs(s - r).max() < 8e-3, np.abs(s - r).max()   # raster resolution 1/400 per axis


def degenerate_scenes():
    """Coincident edges, shared corners, duplicates, containment — the cases where the boundary integral needs its tie rules."""
    out = {}
    out["reference_test"] = np.concatenate([abi.ltwh([0.0], [0.0], [10.0], [10.0]), abi.ltwh([5.0], [5.0], [10.0], [10.0]),
                                            abi.ltwh([10.0], [10.0], [10.0], [10.0])])
    g = [(x * 10.0, y * 10.0) for x in range(4) for y in range(3)]                     # a grid of abutting boxes
    out["grid_abutting"] = abi.ltwh([p[0] for p in g], [p[1] for p in g], [10.0] * 12, [10.0] * 12)
    out["grid_half_overlap"] = abi.ltwh([p[0] / 2 for p in g], [p[1] / 2 for p in g], [10.0] * 12, [10.0] * 12)
    d = abi.ltwh([3.0, 3.0, 3.0, 8.0], [4.0, 4.0, 4.0, 4.0], [10.0, 10.0, 10.0, 10.0], [6.0, 6.0, 6.0, 6.0])
    out["triplicate_plus_shift"] = d
    out["same_left_edge"] = abi.ltwh([0.0, 0.0, 0.0], [0.0, 2.0, 5.0], [10.0, 6.0, 12.0], [10.0, 3.0, 2.0])
    out["nested"] = abi.ltwh([0.0, 2.0, 4.0], [0.0, 2.0, 4.0], [20.0, 10.0, 2.0], [20.0, 10.0, 2.0])
    out["corner_touch"] = abi.ltwh([0.0, 10.0], [0.0, 10.0], [10.0, 10.0], [10.0, 10.0])
    rng = np.random.default_rng(5)
    o = synth.dense_boxes(rng, 12, (300.0, 200.0), oriented=True)
    out["oriented_duplicates"] = np.concatenate([o, o[:6]])
    q = abi.make_boxes([0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [1.0, 1.0, 2.0], [10.0, 10.0, 4.0], angle=[0.3, 0.3 + np.pi / 2, 0.3])
    out["rotated_same_square"] = q   # the same square turned by 90 degrees: every edge coincides with an edge of the other
    return out


@pytest.mark.parametrize("name", sorted(degenerate_scenes()))
def test_emu_degenerate_scenes(name):
    b = degenerate_scenes()[name]
    s, st = emu_shares(b)
    assert st == 0
    assert np.abs(s - O.own_area_shares(b)).max() < EPS, (name, s, O.own_area_shares(b))


@pytest.mark.parametrize("oriented", [False, True])
@pytest.mark.parametrize("n,canvas", [(300, (1920.0, 1080.0)), (120, (300.0, 200.0))])
def test_emu_against_oracle(oriented, n, canvas):
    rng = np.random.default_rng(11 + oriented)
    b = synth.dense_boxes(rng, n, canvas, oriented=oriented)
    s, st = emu_shares(b)
    ref = O.own_area_shares(b)
    assert st == 0
    assert np.abs(s - ref).max() < EPS, np.abs(s - ref).max()


def test_emu_status_bits():
    b = np.repeat(abi.ltwh([0.0], [0.0], [10.0], [10.0]), 6)
    assert emu_shares(b, max_nb=4)[1] & 1
    thin = abi.ltwh(list(np.arange(8) * 3.0), [0.0] * 8, [1.0] * 8, [10.0] * 8)          # 8 disjoint bars cut the long box: 8 disjoint stretches
    long_box = abi.ltwh([-1.0], [4.0], [30.0], [2.0])
    sc = np.concatenate([long_box, thin])
    assert emu_shares(sc, cap=4)[1] & 2
    s, st = emu_shares(sc)
    assert st == 0 and np.abs(s - O.own_area_shares(sc)).max() < EPS


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(degenerate_scenes()))
def test_gpu_degenerate_scenes(name):
    from similari_amd.engine import Engine
    b = degenerate_scenes()[name]
    e = Engine(abi.make_config())
    s = e.own_areas(b)
    e.close()
    assert np.abs(s - O.own_area_shares(b)).max() < EPS, (name, s)


@pytest.mark.gpu
@pytest.mark.parametrize("oriented", [False, True])
@pytest.mark.parametrize("n", [0, 1, 65, 1000])
def test_gpu_against_oracle(oriented, n):
    from similari_amd.engine import Engine
    rng = np.random.default_rng(21 + oriented)
    b = synth.dense_boxes(rng, n, (1920.0, 1080.0), oriented=oriented)
    e = Engine(abi.make_config())
    s = e.own_areas(b)
    assert len(s) == n
    if n:
        ref = O.own_area_shares(b)
        assert np.abs(s - ref).max() < EPS, np.abs(s - ref).max()
        emu, _ = emu_shares(b)
        assert np.abs(s - emu).max() < 1e-6          # same arithmetic, different summation order over the edges
    e.close()


@pytest.mark.gpu
def test_gpu_dense_crowd_takes_the_spill_path():
    """bbox_own_areas.rs:8-46 has no limit on the number of overlapping neighbours; the engine's LDS-resident kernel holds 127 and 24
    disjoint covered stretches per edge, and hands every box beyond that to the spill path (lists in HBM).  200 identical boxes
    (every share 0), a pile of 400 random boxes on a 300 x 300 canvas (150-390 neighbours each), and a long bar cut by 40 posts
    (40 disjoint stretches on one edge) against the oracle's piece-by-piece difference."""
    from similari_amd.engine import Engine
    e = Engine(abi.make_config())
    try:
        many = np.repeat(abi.ltwh([0.0], [0.0], [10.0], [10.0]), 200)
        s = e.own_areas(many)
        assert np.abs(s - O.own_area_shares(many)).max() < EPS and s.max() < EPS
        rng = np.random.default_rng(5)
        pile = synth.dense_boxes(rng, 400, (300.0, 300.0), oriented=True)
        s = e.own_areas(pile)
        ref = O.own_area_shares(pile)
        assert np.abs(s - ref).max() < EPS, np.abs(s - ref).max()
        posts = abi.ltwh(list(np.arange(40) * 3.0), [0.0] * 40, [1.0] * 40, [10.0] * 40)
        bar = abi.ltwh([-1.0], [4.0], [125.0], [2.0])
        sc = np.concatenate([bar, posts])
        s = e.own_areas(sc)
        assert np.abs(s - O.own_area_shares(sc)).max() < EPS
        assert abs(float(s[0]) - (125.0 - 40.0) / 125.0) < 1e-4
    finally:
        e.close()
