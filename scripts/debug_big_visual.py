"""Debug aid: the big VisualSORT frame with a dense positional stage — where do the GPU's ids differ from the oracle's, and is the
total weight of the positional stage the same?"""
import sys
sys.path[:0] = ["/root/repo", "/root/repo/tests", ".", "tests"]
import numpy as np
import oracle_lib as O
from similari_amd import abi, synth
from similari_amd.engine import Engine

def run(n, t, seed=67, canvas=(400.0, 300.0)):
    rng = np.random.default_rng(seed)
    d = 64
    sc = synth.visual_scene(rng, t, n, d, 1, canvas=canvas, new_fraction=0.2)
    sc["det_quality"][rng.uniform(size=n) < 0.15] = 0.05
    cfg = abi.make_config(positional="iou", positional_threshold=0.05, visual="cosine", visual_threshold=0.2, feature_len=d, max_observations=1,
                          visual_min_votes=1, visual_minimal_track_length=1, visual_minimal_quality_use=0.3, positional_min_confidence=0.1, max_idle_epochs=5)
    tracks = abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"], feats=sc["track_feats"], feat_present=sc["track_present"])
    det = abi.make_detections(sc["det_boxes"], feats=sc["det_feats"], feat_quality=sc["det_quality"])
    ref = O.associate(cfg, tracks, 1, det)
    eng = Engine(cfg)
    eng.upsert(0, tracks)
    ids, votes = eng.associate(0, 1, det)
    q = eng.tap_quantised()
    eng.close()
    bad = np.nonzero(ids != ref["track_id"])[0]
    thr_q = 50000
    def gain(idv, vt):
        g = 0
        for i in range(n):
            if vt[i] == abi.SA_VOTE_POSITIONAL and idv[i]:
                g += int(q[i, int(idv[i]) - 1]) - thr_q
        return g
    print(f"n={n} t={t}: mismatching rows {len(bad)}; votes equal {np.array_equal(votes, ref['voting_type'])}; positional gain gpu {gain(ids, votes)} oracle {gain(ref['track_id'], ref['voting_type'])}")
    print("  visual rows equal:", np.array_equal(ids[votes == 1], ref["track_id"][votes == 1]), " n positional gpu/oracle", int((votes == 2).sum()), int((ref["voting_type"] == 2).sum()))
    for i in bad[:12]:
        print("   row", i, "gpu", int(ids[i]), int(votes[i]), "oracle", int(ref["track_id"][i]), int(ref["voting_type"][i]),
              "q gpu", int(q[i, int(ids[i]) - 1]) if ids[i] else None, "q oracle", int(q[i, int(ref['track_id'][i]) - 1]) if ref["track_id"][i] else None)

for n, t in ((1300, 1300), (1100, 1100), (1025, 1025), (600, 600)):
    run(n, t)
