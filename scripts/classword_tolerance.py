"""How often does the DEFAULT BestFit path for banks of 2..8 observations (class words: groups ranked by W = c max_dist - sum w)
decide differently from the exact-formula path (SA_FLAG_BESTFIT_TILE: the reference's sum of f64(f32(max_dist - w_k)), best.rs:93-95) and
from the oracle?  Both GPU paths see the SAME weights bit for bit (same contraction), so a difference between them is the class words'
own tolerance and nothing else.
  realistic:    C2-sized frames, three observations per track (1 M groups each), plain / churned / ragged banks
  adversarial:  pairs of near-duplicate tracks (row decisions) and near-duplicate detections (column decisions) whose group sums differ
                by 0 .. `ulps` f32 ulps of max_dist
Prints one JSON line per set."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np
from similari_amd import abi, synth
from similari_amd.engine import Engine


def run(cfg_kw, sc, flags):
    cfg = abi.make_config(**cfg_kw)
    cfg.flags |= flags
    tracks = abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"], feats=sc["track_feats"], feat_present=sc["track_present"])
    det = abi.make_detections(sc["det_boxes"], feats=sc["det_feats"], feat_quality=sc["det_quality"])
    eng = Engine(cfg)
    try:
        eng.upsert(0, tracks)
        ids, votes = eng.associate(0, 1, det)
    finally:
        eng.close()
    return ids.copy(), votes.copy(), (cfg, tracks, det)


def nudge(rng, a, f, drop):
    """a + s r (r orthogonal to f): the cosine to f falls by `drop`."""
    ca = a @ f / np.sqrt((a @ a) * (f @ f))
    r = rng.standard_normal(len(a))
    r -= (r @ f) / (f @ f) * f
    r *= np.linalg.norm(a) / np.linalg.norm(r)
    lo, hi = 0.0, 1.0
    for _ in range(70):
        mid = 0.5 * (lo + hi)
        b = a + mid * r
        cb = b @ f / np.sqrt((b @ b) * (f @ f))
        lo, hi = (mid, hi) if cb > ca - drop else (lo, mid)
    return a + 0.5 * (lo + hi) * r


def adversarial_scene(rng, pairs=250, d=512, k=3, ulps=4.0):
    ulp = 2.0 ** -23
    ident = synth.reid_identities(rng, 2 * pairs, d).astype(np.float64)
    banks, dets, kinds = [], [], []   # kind of a detection's decision: 0 = row near-tie, 1 = row EXACT tie (permuted bank), 2 = column near-tie
    # rows: detection x sees tracks A and B = A with one observation nudged by 0 .. ulps ulp of its weight's scale
    for j in range(pairs):
        a = np.stack([ident[j] + rng.uniform(-0.01, 0.01, d) for _ in range(k)])
        x = ident[j] + rng.uniform(-0.01, 0.01, d)
        b = a.copy()
        kk = int(rng.integers(0, k))
        # the nudged row moves AWAY from the detection: the detection's weight on it rises by ~ drop
        b[kk] = nudge(rng, a[kk], x, rng.uniform(0.0, ulps) * ulp)
        exact = rng.uniform() < 0.3
        if exact:
            b = a[np.roll(np.arange(k), 1)]      # the same multiset of observations in another order: an exact tie
        banks += [a, b]
        dets.append(x)
        kinds.append(1 if exact else 0)
    # columns: track F is seen by detections x and y = x nudged
    for j in range(pairs, 2 * pairs):
        f = np.stack([ident[j] + rng.uniform(-0.01, 0.01, d) for _ in range(k)])
        x = ident[j] + rng.uniform(-0.01, 0.01, d)
        y = nudge(rng, x, f.mean(0), rng.uniform(0.0, ulps) * ulp)
        banks.append(f)
        dets += [x, y]
        kinds += [2, 2]
    bank = np.stack(banks).astype(np.float32)
    det = np.stack(dets).astype(np.float32)
    t, n = len(bank), len(det)
    perm = rng.permutation(n)
    return dict(track_ids=np.arange(1, t + 1, dtype=np.uint64), track_boxes=synth.dense_boxes(rng, t, (8000.0, 6000.0)),
                track_epochs=np.zeros(t, np.uint64), track_feats=bank, track_present=np.ones((t, k), np.uint8),
                det_boxes=synth.dense_boxes(rng, n, (8000.0, 6000.0)), det_feats=det[perm].copy(), det_quality=np.full(n, 0.9, np.float32), kinds=np.array(kinds)[perm])


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    kw = dict(positional="iou", positional_threshold=0.3, visual="cosine", visual_threshold=0.2, feature_len=512, max_observations=3,
              visual_min_votes=1, visual_minimal_track_length=1, positional_min_confidence=0.1, max_idle_epochs=5)
    import oracle_lib as O
    tot = dict(groups=0, rows=0, differ_default_vs_tile=0, differ_default_vs_oracle=0, differ_tile_vs_oracle=0, oracle_frames=0)
    for s in range(seeds):
        rng = np.random.default_rng(1000 + s)
        sc = synth.visual_scene(rng, 1000, 1000, 512, 3, new_fraction=0.1 if s % 3 == 1 else 0.0)
        if s % 3 == 2:   # ragged banks: a third of the observations missing
            sc["track_present"] = (rng.uniform(size=sc["track_present"].shape) > 0.33).astype(np.uint8)
            sc["track_present"][:, 0] = 1
        ids_d, v_d, (cfg, tracks, det) = run(kw, sc, 0)
        ids_t, v_t, _ = run(kw, sc, abi.SA_FLAG_BESTFIT_TILE)
        tot["groups"] += 1000 * 1000
        tot["rows"] += 1000
        tot["differ_default_vs_tile"] += int(((ids_d != ids_t) | (v_d != v_t)).sum())
        if s < 2:
            ref = O.associate(abi.make_config(**kw), tracks, 1, det)
            tot["oracle_frames"] += 1
            tot["differ_default_vs_oracle"] += int(((ids_d != ref["track_id"]) | (v_d != ref["voting_type"])).sum())
            tot["differ_tile_vs_oracle"] += int(((ids_t != ref["track_id"]) | (v_t != ref["voting_type"])).sum())
    print(json.dumps(dict(set="realistic: C2-sized frames, 3 observations per track", **tot)), flush=True)
    for ulps in (1.0, 4.0, 16.0):
        tot = dict(decisions=0, differ_default_vs_tile=0, differ_default_vs_oracle=0, differ_tile_vs_oracle=0, exact_tie_rows=0,
                   exact_ties_default_vs_oracle=0, exact_ties_tile_vs_oracle=0)
        for s in range(4):
            rng = np.random.default_rng(2000 + s)
            sc = adversarial_scene(rng, ulps=ulps)
            ids_d, v_d, (cfg, tracks, det) = run(kw, sc, 0)
            ids_t, v_t, _ = run(kw, sc, abi.SA_FLAG_BESTFIT_TILE)
            ref = O.associate(abi.make_config(**kw), tracks, 1, det)
            tot["decisions"] += len(ids_d)
            tot["differ_default_vs_tile"] += int(((ids_d != ids_t) | (v_d != v_t)).sum())
            tot["differ_default_vs_oracle"] += int(((ids_d != ref["track_id"]) | (v_d != ref["voting_type"])).sum())
            tot["differ_tile_vs_oracle"] += int(((ids_t != ref["track_id"]) | (v_t != ref["voting_type"])).sum())
            ex = sc["kinds"] == 1
            tot["exact_tie_rows"] += int(ex.sum())
            tot["exact_ties_default_vs_oracle"] += int(((ids_d != ref["track_id"]) & ex).sum())
            tot["exact_ties_tile_vs_oracle"] += int(((ids_t != ref["track_id"]) & ex).sum())
        print(json.dumps(dict(set=f"adversarial: group sums within 0..{ulps:g} ulp of max_dist (row and column near-ties, 30 % exact ties by permuted banks)", **tot)), flush=True)


if __name__ == "__main__":
    main()
