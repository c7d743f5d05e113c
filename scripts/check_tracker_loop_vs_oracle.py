"""The churned tracker loop of scripts/bench_tracker.py (1000 objects, 5 % leave / enter per frame, max_idle_epochs 3) through the
facade with device upkeep AND through the oracle tracker, frame by frame: ids, lengths, vote types, boxes.  A full-size check of
everything the loop exercises at once — the many-workgroup tail beyond 1024 tracks, eviction from the engine's table, the deferred
half of the bookkeeping, the upkeep queued behind the association.   python scripts/check_tracker_loop_vs_oracle.py [sort|visual] [frames]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import oracle_lib as O  # noqa: E402
from similari_amd import synth  # noqa: E402
from similari_amd import trackers as TR  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "visual"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n, d, churn = 1000, 512, 0.05
rng = np.random.default_rng(0)
pool = n + frames * (n // 20) + 8
ident = synth.reid_identities(rng, pool, d)
world = synth.dense_boxes(rng, pool, (1920.0, 1080.0))
if kind == "visual":
    opts = (TR.VisualSortOptions().max_idle_epochs(3).kept_history_length(3).visual_metric(TR.VisualSortMetricType.cosine(0.2))
            .positional_metric(TR.PositionalMetricType.iou(0.3)).visual_minimal_track_length(1).visual_max_observations(3))
    og, keep_g = TR.visual_options(opts, d, False, -1, True)
    oo, keep_o = TR.visual_options(opts, d, False, -1, False)
else:
    og, keep_g = TR.sort_options(3, 3, TR.PositionalMetricType.iou(0.3), 0.05, None, 1.0 / 20.0, 1.0 / 160.0, batch=False, device_upkeep=True)
    oo, keep_o = TR.sort_options(3, 3, TR.PositionalMetricType.iou(0.3), 0.05, None, 1.0 / 20.0, 1.0 / 160.0, batch=False, device_upkeep=False)
g, o = TR._Tracker(og, keep_g), O.OracleTracker(oo, keep_o)
r = np.random.default_rng(1)
active = np.arange(n)
fresh = n
for f in range(frames):
    world = synth.jitter_boxes(r, world, 2.0)
    if f > 0:
        k_out = max(1, int(churn * n))
        gone = r.choice(n, k_out, replace=False)
        active[gone] = np.arange(fresh, fresh + k_out)
        fresh += k_out
    boxes = [TR.Universal2DBox(float(b["xc"]), float(b["yc"]), None, float(b["aspect"]), float(b["height"]), float(b["confidence"])) for b in world[active]]
    if kind == "visual":
        feats = synth.observe(r, ident[active], 0.01)
        items = [TR.VisualSortObservation(feats[k], 0.9, boxes[k], None) for k in range(n)]
    else:
        items = [(boxes[k], None) for k in range(n)]
    rg, ro = g.predict(items), o.predict(items)
    bad = [k for k in range(n) if (rg[k].id, rg[k].length, rg[k].voting_type, rg[k].epoch) != (ro[k].id, ro[k].length, ro[k].voting_type, ro[k].epoch)
           or (rg[k].predicted_bbox.xc, rg[k].predicted_bbox.yc, rg[k].predicted_bbox.height) != (ro[k].predicted_bbox.xc, ro[k].predicted_bbox.yc, ro[k].predicted_bbox.height)]
    new = sum(1 for x in rg if x.length == 1)
    print(f"frame {f}: {new} tracks started, {len(bad)} of {n} differ from the oracle tracker", flush=True)
    assert not bad, bad[:10]
assert g.active_tracks() == o.active_tracks()
print("LOOP-OK", kind, frames)
