"""Non-maximum suppression (src/utils/nms.rs:32-72).

cpu : the oracle's or_nms on the reference's own examples — the (disabled) unit test of nms.rs:142-156 and the doc example of
      utils/nms/nms_py.rs:24-37 — and on properties that follow from the algorithm.
gpu : sa_nms through the C ABI returns exactly the oracle's indices, in the same order, for axis-aligned and oriented boxes,
      with and without scores, on dense random scenes up to the reference bench size (benches/nms.rs: 1000 boxes)."""
import numpy as np
import pytest

import oracle_lib as O
from similari_amd import abi, synth


def xyaah(xc, yc, aspect, h, angle=None):
    return abi.make_boxes([xc], [yc], [aspect], [h], angle=None if angle is None else [angle])[0]


def test_reference_unit_example():
    # nms.rs:145-151: three concentric boxes and one apart, threshold 0.8, no scores -> rank = height
    b = np.array([xyaah(0, 0, 1.0, 5.0), xyaah(0, 0, 1.05, 5.1), xyaah(0, 0, 1.0, 4.9), xyaah(3, 4, 1.0, 4.5)], abi.BOX_DTYPE)
    keep = O.nms(b, None, 0.8, None)
    # visited by height: #1 (5.1) suppresses #0 (25 / 25 = 1.0) and #2 (24.01 / 24.01); #3 overlaps #1 by 2.85 x 1.35 / 20.25 = 0.19
    assert keep.tolist() == [1, 3]


def test_reference_doc_example():
    # nms_py.rs:24-37 — ltwh boxes (10, 11, 3, 3.8) score 1.0 and (10.3, 11.1, 2.9, 3.9) score 0.9; the second is suppressed
    b = np.concatenate([abi.ltwh([10.3], [11.1], [2.9], [3.9]), abi.ltwh([10.0], [11.0], [3.0], [3.8])])
    assert O.nms(b, [0.9, 1.0], 0.7, 0.0).tolist() == [1]
    # without scores the taller box ranks first (3.9 vs 4.0 in the doc's second call)
    b2 = np.concatenate([abi.ltwh([10.3], [11.1], [2.9], [3.9]), abi.ltwh([10.0], [11.0], [3.0], [4.0])])
    assert O.nms(b2, None, 0.7, 0.0).tolist() == [1]


def test_filters_and_order():
    rng = np.random.default_rng(0)
    b = synth.dense_boxes(rng, 40, (4000.0, 4000.0))          # far apart: nothing is suppressed
    s = rng.uniform(0, 1, 40).astype(np.float32)
    keep = O.nms(b, s, 0.5, 0.3)
    assert set(keep.tolist()) == set(np.nonzero(s > 0.3)[0].tolist())
    assert (np.diff(s[keep]) <= 0).all(), "output is in descending rank"
    # equal ranks keep their input order (stable sort)
    s2 = np.full(40, 0.5, np.float32)
    assert O.nms(b, s2, 0.5, None).tolist() == list(range(40))
    # NaN score = None: passes any threshold, ranks by height
    s3 = s.copy(); s3[::3] = np.nan
    k3 = O.nms(b, s3, 0.5, 0.99)
    assert set(k3.tolist()) >= set(range(0, 40, 3))


def scene(rng, n, oriented):
    b = synth.dense_boxes(rng, n, (1200.0, 800.0), oriented=oriented)
    # clusters of near-duplicates, as a detector emits them
    dup = synth.jitter_boxes(rng, b[: n // 2], 3.0, size_rel=0.05, angle_sigma=0.05 if oriented else 0.0)
    return np.concatenate([b, dup])[rng.permutation(n + n // 2)]


@pytest.mark.gpu
@pytest.mark.parametrize("oriented", [False, True])
@pytest.mark.parametrize("n", [1, 17, 200, 1000])
@pytest.mark.parametrize("with_scores", [False, True])
def test_gpu_nms_matches_oracle(oriented, n, with_scores):
    from similari_amd.engine import Engine

    rng = np.random.default_rng(1000 * oriented + n + with_scores)
    b = scene(rng, n, oriented)
    s = rng.uniform(0, 1, len(b)).astype(np.float32) if with_scores else None
    if with_scores:
        s[::7] = np.nan
    eng = Engine(abi.make_config())
    try:
        for thr, sthr in ((0.5, None), (0.3, 0.2), (0.8, None), (-1.0, None)):
            got = eng.nms(b, s, thr, sthr)
            want = O.nms(b, s, thr, sthr)
            np.testing.assert_array_equal(got, want)
        if n >= 200:
            assert len(want) < len(b), "the duplicates must be suppressed at some threshold"
    finally:
        eng.close()


@pytest.mark.gpu
def test_gpu_nms_edge_cases():
    from similari_amd.engine import Engine

    eng = Engine(abi.make_config())
    try:
        assert eng.nms(np.zeros(0, abi.BOX_DTYPE)).tolist() == []
        b = np.array([xyaah(0, 0, 1.0, 5.0), xyaah(0, 0, 1.0, 5.0)], abi.BOX_DTYPE)
        assert eng.nms(b, None, 0.5, None).tolist() == [0]           # identical boxes: the first one stays
        assert eng.nms(b, None, 1.0, None).tolist() == [0, 1]        # metric 1.0 is not > 1.0
        assert eng.nms(b, [0.1, 0.9], 0.5, 0.5).tolist() == [1]
        assert eng.nms(b, [0.1, 0.2], 0.5, 0.5).tolist() == []
    finally:
        eng.close()
