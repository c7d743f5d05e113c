"""The general tail's register path for two-row components (k_assign_solve, sa_kernels.hip) restated in Python against brute force
— no GPU.  Two rows, at most four usable records each (gain > 0, column): each row bids for its heaviest record (lowest column on
ties); different columns -> both keep their bids; the same column -> the better of (root keeps it, the other row takes its best
record on another column) and the converse, the root keeping it on a tie.  The total must be the optimum of the 2 x (columns + self)
assignment problem the reference solves with kuhn_munkres (sort/voting.rs:44-86: an unmatched row falls back to its self column,
gain 0), and where the optimum is unique the matching itself."""
import itertools

import numpy as np


def best_of(edges, not_col):
    at, bg = -1, 0
    for k, (g, c) in enumerate(edges):
        if c == not_col:
            continue
        if at < 0 or g > bg or (g == bg and c < edges[at][1]):
            at, bg = k, g
    return (bg if at >= 0 else 0), at


def pair_path(a, b):
    ga, ia = best_of(a, -1)
    gb, ib = best_of(b, -1)
    if ia >= 0 and ib >= 0 and a[ia][1] == b[ib][1]:
        ga2, ia2 = best_of(a, a[ia][1])
        gb2, ib2 = best_of(b, b[ib][1])
        if ga + gb2 >= ga2 + gb:
            ib = ib2
        else:
            ia = ia2
    return (a[ia][1] if ia >= 0 else -1), (b[ib][1] if ib >= 0 else -1)


def brute(a, b):
    best, sols = -1, []
    for x, y in itertools.product([None] + a, [None] + b):
        if x is not None and y is not None and x[1] == y[1]:
            continue
        tot = (x[0] if x else 0) + (y[0] if y else 0)
        sol = (x[1] if x else -1, y[1] if y else -1)
        if tot > best:
            best, sols = tot, [sol]
        elif tot == best and sol not in sols:
            sols.append(sol)
    return best, sols


def test_pair_path_is_optimal():
    rng = np.random.default_rng(5)
    unique = 0
    for trial in range(20000):
        ncols = int(rng.integers(1, 7))
        def row():
            n = int(rng.integers(1, 5))
            cols = rng.choice(ncols + 3, size=min(n, ncols + 3), replace=False)
            hi = 6 if trial % 3 == 0 else 1_000_000        # every third trial: small integer gains -> plenty of ties
            return [(int(rng.integers(1, hi + 1)), int(c)) for c in cols]
        a, b = row(), row()
        ca, cb = pair_path(a, b)
        assert ca < 0 or cb < 0 or ca != cb
        gain = dict((c, g) for g, c in a).get(ca, 0) + dict((c, g) for g, c in b).get(cb, 0)
        best, sols = brute(a, b)
        assert gain == best, (a, b, ca, cb)
        if len(sols) == 1:
            unique += 1
            assert (ca, cb) == sols[0]
    assert unique > 10000
