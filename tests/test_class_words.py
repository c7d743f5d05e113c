"""The count-class reduction of the BestFit vote (deeper banks without the weight matrix: visual_ktile in sa_gemm.hip, the SCN_WORDSK
branch of k_assign_small) restated in numpy against the reference's formula — no GPU.

Reference (track/voting/best.rs:59-100, the oracle's bestfit_voting): a group (candidate, track) with c present observations weighs
W = sum_k f64(f32(max_dist - w_k)), max_dist = the frame's largest present weight; a row's (track's) best group is the heaviest, the
lowest index among equals.  Product: per count class c the group with the SMALLEST sum of weights (as the order-preserving key of
the f32 sum, lowest index on ties), then W_c = c max_dist - sum across the classes.  The two can only differ where two groups' weights
are within rounding of each other: c/2 ulp of max_dist from the f32 differences plus half an ulp of the f32 sum."""
import numpy as np
import pytest


def reference_best(w, min_votes):
    present = ~np.isnan(w)
    max_dist = np.float32(np.nanmax(w))
    diff = (max_dist - w).astype(np.float32).astype(np.float64)
    cnt = present.sum(axis=2)
    W = np.where(present, diff, 0.0).sum(axis=2)
    ok = (cnt >= 1) & (cnt >= min_votes)
    return np.where(ok, W, -np.inf), max_dist


def class_words_best(w, min_votes, axis):
    """argmax over `axis` (1: a candidate's best track, 0: a track's best candidate) the way the tiles + the tail find it."""
    present = ~np.isnan(w)
    max_dist = np.float64(np.float32(np.nanmax(w)))
    k = w.shape[2]
    cnt = present.sum(axis=2)
    s = np.where(present, w.astype(np.float64), 0.0).sum(axis=2).astype(np.float32)  # what the key of the f32 sum keeps
    n = w.shape[0] if axis == 1 else w.shape[1]
    best_idx = np.full(n, -1)
    best_w = np.full(n, -np.inf)
    for c in range(max(1, min_votes), k + 1):
        m = np.where(cnt == c, s, np.inf)
        m = m if axis == 1 else m.T
        idx = m.argmin(axis=1)                      # lowest index among equal keys: the 64-bit minimum of (key << 32 | index)
        val = m[np.arange(n), idx]
        wc = c * max_dist - val.astype(np.float64)
        take = np.isfinite(val) & ((wc > best_w) | ((wc == best_w) & (idx < best_idx)))
        best_idx = np.where(take, idx, best_idx)
        best_w = np.where(take, wc, best_w)
    return best_idx, best_w


@pytest.mark.parametrize("k,min_votes", [(2, 1), (3, 1), (3, 2), (5, 1), (5, 3), (8, 1)])
def test_class_reduction_finds_the_reference_winner(k, min_votes):
    rng = np.random.default_rng(100 * k + min_votes)
    worst = 0.0
    for trial in range(30):
        n, t = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        w = rng.uniform(0.05, 1.2, (n, t, k)).astype(np.float32)
        w[rng.uniform(size=w.shape) < rng.uniform(0.0, 0.6)] = np.nan
        if np.isnan(w).all():
            continue
        W, max_dist = reference_best(w, min_votes)
        tol = (k + 1) * float(np.spacing(np.float32(max(1.0, max_dist * k))))      # k half-ulps of the differences + the sum's own
        for axis in (1, 0):
            m = W if axis == 1 else W.T
            ref_idx = m.argmax(axis=1)
            ref_w = m[np.arange(m.shape[0]), ref_idx]
            got_idx, got_w = class_words_best(w, min_votes, axis)
            none = ~np.isfinite(ref_w)
            assert (got_idx[none] == -1).all()
            rows = np.where(~none)[0]
            assert (got_idx[rows] >= 0).all()
            # the chosen group is the reference's, or weighs within rounding of it; the weight the tail compares is within rounding too
            chosen = m[rows, got_idx[rows]]
            assert (ref_w[rows] - chosen <= tol).all()
            assert (np.abs(got_w[rows] - chosen) <= tol).all()
            worst = max(worst, float((ref_w[rows] - chosen).max()))
            far = rows[(ref_w[rows] - np.partition(np.where(np.isfinite(m[rows]), m[rows], -1e30), -2, axis=1)[:, -2] > 4 * tol) if m.shape[1] > 1 else np.ones(len(rows), bool)]
            np.testing.assert_array_equal(got_idx[far], ref_idx[far])  # a clear runner-up: the very same group
    assert worst <= 1e-5
