"""The N > 1 path on CPU: world_size-2 `gloo` run of similari_amd.sharding (scene scatter -> per-rank batched tracker ->
result gather).  The per-rank tracker here is the oracle's batched tracker loop (tests may use the oracle; the sharding layer
itself computes nothing), so the test pins the distribution logic: ownership, wire format, ordering, id namespacing.

Expected values: for every rank r, a fresh single-process tracker fed ONLY the scenes r owns, frame by frame — scenes never
interact (compatible() is false across scene ids, sort.rs:251), so sharding must not change any per-scene result."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
SCENES = (3, 4, 7, 10, 11)
FRAMES = 5
WORLD = 2


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def make_requests(visual: bool, seed: int):
    """Seeded multi-scene sequence; returns a list of {scene: [items]} per frame (plain python, picklable)."""
    sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
    from similari_amd import synth
    from similari_amd import trackers as TR

    rng = np.random.default_rng(seed)
    n, d = 12, 16
    world = {s: synth.dense_boxes(rng, n, (600.0, 400.0), oriented=(s % 2 == 1)) for s in SCENES}
    ident = {s: synth.reid_identities(rng, n, d) for s in SCENES}
    frames = []
    for f in range(FRAMES):
        req = {}
        for s in SCENES:
            if f == 2 and s == 7:
                continue  # a scene may be absent from a batch
            world[s] = synth.jitter_boxes(rng, world[s], 1.5, angle_sigma=0.01 if s % 2 == 1 else 0.0)
            order = rng.permutation(n)[: n - int(rng.integers(0, 3))]
            items = []
            for k in order:
                b = world[s][k]
                box = TR.Universal2DBox(float(b["xc"]), float(b["yc"]), float(b["angle"]) if b["has_angle"] else None, float(b["aspect"]),
                                        float(b["height"]), float(b["confidence"]))
                cid = int(k) if k % 3 == 0 else None
                if visual:
                    ft = None if k % 5 == 4 else synth.observe(rng, ident[s][k:k + 1], 0.01)[0]
                    items.append(TR.VisualSortObservation(ft, None if k % 4 == 0 else float(rng.uniform(0.5, 1.0)), box, cid))
                else:
                    items.append((box, cid))
            req[s] = items
        frames.append(req)
    return frames


def make_local_tracker(visual: bool):
    import oracle_lib as O
    from similari_amd import trackers as TR

    if visual:
        opts = (TR.VisualSortOptions().max_idle_epochs(3).kept_history_length(3).visual_metric(TR.VisualSortMetricType.cosine(0.5))
                .positional_metric(TR.PositionalMetricType.iou(0.3)).visual_minimal_track_length(1).visual_max_observations(3))
        o, keep = TR.visual_options(opts, 16, batch=True)
    else:
        o, keep = TR.sort_options(3, 3, TR.PositionalMetricType.iou(0.3), 0.05, None, 1.0 / 20.0, 1.0 / 160.0, batch=True)
    return O.OracleTracker(o, keep)


def rows(tracks):
    from golden import make_golden as G

    return G.track_rows(tracks)


def worker(rank: int, port: int, visual: bool, outfile: str):
    sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    import torch.distributed as dist

    from similari_amd import sharding
    from similari_amd import trackers as TR

    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        sh = sharding.ShardedBatchTracker(make_local_tracker(visual), feature_len=16 if visual else 0)
        if rank != 0:
            sh.serve_forever()
            return
        out = {}
        for f, req in enumerate(make_requests(visual, seed=5 + visual)):
            batch = TR.PredictionBatchRequest()
            batch.scenes = req
            res = sh.predict(batch)
            assert list(res.keys()) == list(req.keys())
            for s, tracks in res.items():
                out[f"f{f}_s{s}"] = rows(tracks)
        sh.shutdown()
        np.savez(outfile, **out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("visual", [False, True], ids=["batch_sort", "batch_visual_sort"])
def test_two_rank_gloo_sharding_matches_per_scene_runs(visual, tmp_path):
    import torch.multiprocessing as mp

    from similari_amd import sharding
    from similari_amd import trackers as TR

    outfile = str(tmp_path / "sharded.npz")
    mp.spawn(worker, args=(free_port(), visual, outfile), nprocs=WORLD, join=True)
    got = dict(np.load(outfile))

    frames = make_requests(visual, seed=5 + visual)
    seen_ids = set()
    for r in range(WORLD):
        trk = make_local_tracker(visual)  # what rank r's tracker must have seen: only its own scenes, in request order
        for f, req in enumerate(frames):
            sub = TR.PredictionBatchRequest()
            sub.scenes = {s: v for s, v in req.items() if sharding.owner(s, WORLD) == r}
            if not sub.scenes:
                continue
            want = trk.predict_batch(sub)
            for s, tracks in want.items():
                w = rows(tracks)
                w[:, 0] = [sharding.global_id(int(i), r, WORLD) for i in w[:, 0]]
                g = got[f"f{f}_s{s}"]
                np.testing.assert_array_equal(np.nan_to_num(g, nan=-7.0), np.nan_to_num(w, nan=-7.0), err_msg=f"frame {f} scene {s}")
                seen_ids.update((int(i), s) for i in g[:, 0])
        trk.close()
    # ids are globally unique: an id never shows up in two scenes
    by_id = {}
    for i, s in seen_ids:
        assert by_id.setdefault(i, s) == s, f"track id {i} appears in scenes {by_id[i]} and {s}"
    assert len(got) == sum(len(fr) for fr in frames)


def test_wire_roundtrip_and_partition():
    sys.path[:0] = [str(ROOT / "tests")]
    from similari_amd import sharding
    from similari_amd import trackers as TR

    req = make_requests(True, seed=1)[0]
    enc = sharding.encode_request(req, 16)
    dec = sharding.decode_request(enc)
    assert list(dec.scenes.keys()) == list(req.keys())
    for s in req:
        for a, b in zip(req[s], dec.scenes[s]):
            assert (a.feature is None) == (b.feature is None)
            if a.feature is not None:
                np.testing.assert_array_equal(a.feature, b.feature)
            assert a.custom_object_id == b.custom_object_id and a.feature_quality == pytest.approx(b.feature_quality, rel=1e-7, nan_ok=True) \
                if a.feature_quality is not None else b.feature_quality is None
            assert np.float32(a.bounding_box.xc) == np.float32(b.bounding_box.xc) and a.bounding_box.angle == b.bounding_box.angle
    batch = TR.PredictionBatchRequest()
    batch.scenes = req
    parts = sharding.partition(batch, 4)
    assert sorted(s for p in parts for s in p) == sorted(req) and all(s % 4 == r for r, p in enumerate(parts) for s in p)
    assert sharding.global_id(1, 0, 1) == 1 and sharding.global_id(5, 3, 8) == 36 and sharding.global_id(0, 3, 8) == 0
    # empty request encodes and decodes
    assert sharding.decode_request(sharding.encode_request({}, 0)).scenes == {}
