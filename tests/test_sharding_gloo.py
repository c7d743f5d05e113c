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


# ---- array-level sharding (ShardedAssociator): the association itself, scattered as arrays ------------------------------------
class OracleEngine:
    """Engine-shaped shim over the oracle for the CPU run (the sharding layer computes nothing; on a GPU box the same class wraps
    similari_amd.engine.Engine — see tests/test_gpu_cluster.py)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.tracks = {}

    def upsert(self, scene, tracks):
        self.tracks[scene] = tracks

    def associate_batch(self, req, res, n=None):
        import ctypes as C

        import oracle_lib as O

        for i in range(len(req) if n is None else n):
            r = O.associate(self.cfg, self.tracks[req[i].scene_id], req[i].epoch, req[i].detections, want_matrices=False)
            m = req[i].detections.n
            np.ctypeslib.as_array(res[i].out_track_id, (m,))[:] = r["track_id"] if m else []
            np.ctypeslib.as_array(res[i].out_voting_type, (m,))[:] = r["voting_type"] if m else []


def assoc_scenes(seed):
    sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
    from similari_amd import abi, synth

    rng = np.random.default_rng(seed)
    d = 24
    cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual="cosine", visual_threshold=0.3, feature_len=d, max_observations=2,
                          visual_min_votes=1, visual_minimal_track_length=1, max_idle_epochs=5)
    scenes = {}
    for s, (n, t) in zip((2, 5, 8, 9, 12), ((30, 40), (0, 10), (25, 25), (41, 33), (17, 60))):
        scenes[s] = synth.visual_scene(rng, t, n, d, 2, canvas=(700.0, 500.0), new_fraction=0.15)
    return cfg, scenes


def assoc_worker(rank: int, port: int, outfile: str):
    sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    import torch.distributed as dist

    from similari_amd import sharding

    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        cfg, scenes = assoc_scenes(9)
        sh = sharding.ShardedAssociator(OracleEngine(cfg), capacity_bytes=1 << 16, capacity_rows=256)
        for s, sc in scenes.items():  # collective: every rank takes part, the owner keeps the rows
            sh.upsert_arrays(s, **(dict(ids=sc["track_ids"], boxes=sc["track_boxes"], epochs=sc["track_epochs"], feats=sc["track_feats"],
                                        feat_present=sc["track_present"]) if rank == 0 else {}))
        if rank != 0:
            sh.serve_forever()
            return
        out = {}
        for f in range(3):  # the same scenes in another request order every time, one scene absent from one batch
            order = [s for s in (list(scenes) if f != 1 else list(scenes)[::-1]) if not (f == 2 and s == 8)]
            items = [(s, 1, scenes[s]["det_boxes"], scenes[s]["det_feats"], scenes[s]["det_quality"]) for s in order]
            if f == 1:
                # a request set beyond the agreed capacity is refused on EVERY rank (the root checks before the first collective and the
                # verdict travels with the shares): the root gets the error, the workers skip the set and keep serving — nobody hangs
                big = [(s, 1, np.concatenate([scenes[s]["det_boxes"]] * 12), None, None) for s in order]
                try:
                    sh.associate(big)
                    raise AssertionError("an oversized request set went through")
                except RuntimeError as ex:
                    assert "refused" in str(ex) and "capacity_rows" in str(ex)
            res = sh.associate(items)
            assert len(res) == len(items)
            for s, (ids, votes) in zip(order, res):
                out[f"f{f}_s{s}_ids"], out[f"f{f}_s{s}_votes"] = ids, votes
        sh.shutdown()
        np.savez(outfile, **out)
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_array_sharding_matches_the_oracle_per_scene(tmp_path):
    import torch.multiprocessing as mp

    import oracle_lib as O
    from similari_amd import abi

    outfile = str(tmp_path / "assoc.npz")
    mp.spawn(assoc_worker, args=(free_port(), outfile), nprocs=WORLD, join=True)
    got = dict(np.load(outfile))
    cfg, scenes = assoc_scenes(9)
    assert len(got) == 2 * (5 + 5 + 4)
    for s, sc in scenes.items():
        tr = abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"], feats=sc["track_feats"], feat_present=sc["track_present"])
        det = abi.make_detections(sc["det_boxes"], feats=sc["det_feats"], feat_quality=sc["det_quality"])
        ref = O.associate(cfg, tr, 1, det, want_matrices=False)
        for f in range(3):
            if f == 2 and s == 8:
                continue
            np.testing.assert_array_equal(got[f"f{f}_s{s}_ids"], ref["track_id"], err_msg=f"frame {f} scene {s}")
            np.testing.assert_array_equal(got[f"f{f}_s{s}_votes"], ref["voting_type"])


class FailingEngine(OracleEngine):
    """Raises inside the share that holds scene 9 (what an engine error — a HIP failure, a refused box — looks like to the layer)."""

    def associate_batch(self, req, res, n=None):
        for i in range(len(req) if n is None else n):
            if req[i].scene_id == 9:
                raise RuntimeError("engine failure inside the share")
        return super().associate_batch(req, res, n)


def fail_worker(rank: int, port: int, outfile: str):
    sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    import torch.distributed as dist

    from similari_amd import sharding

    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        cfg, scenes = assoc_scenes(9)
        sh = sharding.ShardedAssociator(FailingEngine(cfg), capacity_bytes=1 << 16, capacity_rows=256)
        for s, sc in scenes.items():
            sh.upsert_arrays(s, **(dict(ids=sc["track_ids"], boxes=sc["track_boxes"], epochs=sc["track_epochs"], feats=sc["track_feats"],
                                        feat_present=sc["track_present"]) if rank == 0 else {}))
        if rank != 0:
            # scene 9 lives on rank 1 (9 % 2): the first set goes through, the second one fails HERE — after the scatter.  The worker
            # must still reach the gather (the root is waiting in it) and only then raise its own exception out of the loop.
            try:
                sh.serve_forever()
                verdict = "returned"
            except sharding.RequestRefused:
                verdict = "refused"
            except RuntimeError as ex:
                verdict = "raised: " + str(ex)
            np.savez(outfile + ".worker.npz", verdict=np.array(verdict))
            return
        ok_items = [(s, 1, scenes[s]["det_boxes"], scenes[s]["det_feats"], scenes[s]["det_quality"]) for s in (2, 8, 5)]
        res = sh.associate(ok_items)
        assert len(res) == 3
        bad_items = [(s, 1, scenes[s]["det_boxes"], scenes[s]["det_feats"], scenes[s]["det_quality"]) for s in (2, 9)]
        try:
            sh.associate(bad_items)
            verdict = "went through"
        except sharding.ShardFailed as ex:
            verdict = "ShardFailed: " + str(ex)
        np.savez(outfile, verdict=np.array(verdict))
        sh.close()  # (no shutdown broadcast: the failed worker has left its loop)
    finally:
        dist.destroy_process_group()


def test_a_rank_that_fails_inside_its_share_still_reaches_the_gather(tmp_path):
    """a worker whose engine call fails between the scatter and the gather used to swallow the error and loop back
    into the next scatter — the root then hung in the gather.  Now the rank takes part in the gather with an error marker, the root
    raises ShardFailed, and the worker's own exception ends its loop (only RequestRefused is skipped)."""
    import torch.multiprocessing as mp

    outfile = str(tmp_path / "fail.npz")
    mp.spawn(fail_worker, args=(free_port(), outfile), nprocs=WORLD, join=True)
    root = str(np.load(outfile)["verdict"])
    worker = str(np.load(outfile + ".worker.npz")["verdict"])
    assert root.startswith("ShardFailed") and "[1]" in root, root
    assert worker == "raised: engine failure inside the share", worker


def fixed_set_worker(rank: int, port: int, outfile: str):
    """bench.py's --gpus N form on a scene-set workload: a FIXED set of scenes, scene s owned by rank s % N (bench.scene_ids_of_rank),
    every rank generating and seeding exactly its own scenes, rank 0 scattering the whole set and gathering every scene's answer."""
    sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    import torch.distributed as dist

    import bench
    from similari_amd import abi, sharding, synth

    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        total, d = 7, 16
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual="cosine", visual_threshold=0.3, feature_len=d, max_observations=1,
                              visual_min_votes=1, visual_minimal_track_length=1, max_idle_epochs=5)
        gen = lambda sid: synth.visual_scene(np.random.default_rng(1000 * 77 + sid), 30 + sid, 25 + 2 * sid, d, 1, canvas=(600.0, 400.0), new_fraction=0.1)  # noqa: E731
        mine = bench.scene_ids_of_rank(total, WORLD, rank)
        assert all(sharding.owner(s, WORLD) == rank for s in mine)
        eng = OracleEngine(cfg)
        n_mine = (total + WORLD - 1) // WORLD
        sh = sharding.ShardedAssociator(eng, capacity_bytes=n_mine * 64 * 4 * d, capacity_rows=n_mine * 64, max_scenes=n_mine)
        for sid in mine:   # every rank seeds ITS scenes itself (no collective): the set is sticky
            sc = gen(sid)
            eng.upsert(sid, abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"], feats=sc["track_feats"], feat_present=sc["track_present"]))
        dist.barrier()
        if rank != 0:
            sh.serve_forever()
            return
        everyone = [(sid, gen(sid)) for sid in range(total)]
        items = [(sid, 1, sc["det_boxes"], sc["det_feats"], sc["det_quality"]) for sid, sc in everyone]
        out = {}
        for step in range(2):
            res = sh.associate(items)
            for (sid, _), (ids, votes) in zip(everyone, res):
                out[f"t{step}_s{sid}_ids"], out[f"t{step}_s{sid}_votes"] = ids, votes
        sh.shutdown()
        np.savez(outfile, **out)
    finally:
        dist.destroy_process_group()


def test_fixed_scene_set_split_over_two_ranks_matches_the_oracle(tmp_path):
    """The strong-scaling form of bench.py (--gpus N on c2b / c3): 7 scenes split scene_id % 2, two steps; every scene's ids and vote
    types equal the oracle's whichever rank served it."""
    import torch.multiprocessing as mp

    import bench
    import oracle_lib as O
    from similari_amd import abi, synth

    assert bench.scene_ids_of_rank(7, 2, 0) == [0, 2, 4, 6] and bench.scene_ids_of_rank(7, 2, 1) == [1, 3, 5]
    assert sorted(bench.scene_ids_of_rank(64, 8, 3)) == list(range(3, 64, 8))
    # a rank that generates only its own scenes gets the very scenes rank 0 generates for the whole set
    c_all = bench.workload("c3", seed=5, scene_ids=[0, 1, 2, 3])[1]
    c_mine = bench.workload("c3", seed=5, scene_ids=[1, 3])[1]
    np.testing.assert_array_equal(c_all[1]["det_boxes"], c_mine[0]["det_boxes"])
    np.testing.assert_array_equal(c_all[3]["track_boxes"], c_mine[1]["track_boxes"])
    outfile = str(tmp_path / "fixed.npz")
    mp.spawn(fixed_set_worker, args=(free_port(), outfile), nprocs=WORLD, join=True)
    got = dict(np.load(outfile))
    d = 16
    cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual="cosine", visual_threshold=0.3, feature_len=d, max_observations=1,
                          visual_min_votes=1, visual_minimal_track_length=1, max_idle_epochs=5)
    for sid in range(7):
        sc = synth.visual_scene(np.random.default_rng(1000 * 77 + sid), 30 + sid, 25 + 2 * sid, d, 1, canvas=(600.0, 400.0), new_fraction=0.1)
        tr = abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"], feats=sc["track_feats"], feat_present=sc["track_present"])
        det = abi.make_detections(sc["det_boxes"], feats=sc["det_feats"], feat_quality=sc["det_quality"])
        ref = O.associate(cfg, tr, 1, det, want_matrices=False)
        for step in range(2):
            np.testing.assert_array_equal(got[f"t{step}_s{sid}_ids"], ref["track_id"], err_msg=f"scene {sid} step {step}")
            np.testing.assert_array_equal(got[f"t{step}_s{sid}_votes"], ref["voting_type"])


def test_share_pack_roundtrip():
    sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
    from similari_amd import sharding

    cfg, scenes = assoc_scenes(3)
    items = [(s, 7 + s, sc["det_boxes"], sc["det_feats"] if s != 9 else None, sc["det_quality"] if s != 12 else None) for s, sc in scenes.items()]
    back, flags = sharding.unpack_share(sharding.pack_share(items, 24))
    assert flags == 0 and len(back) == len(items)
    for a, b in zip(items, back):
        assert a[0] == b[0] and a[1] == b[1]
        np.testing.assert_array_equal(a[2], b[2])
        for x, y in ((a[3], b[3]), (a[4], b[4])):
            assert (x is None) == (y is None)
            if x is not None:
                np.testing.assert_array_equal(x, y)
    assert sharding.unpack_share(sharding.pack_share([], 0)) == ([], 0)
    # the in-place form (ShardedAssociator writes a share straight into its pinned staging row): the same bytes
    for base in (0, sharding.prefix_bytes(8, sum(len(it[2]) for it in items))):
        ref = sharding.pack_share(items, 24, base, 0)
        dst = np.full(len(ref) + 64, 0xAB, np.uint8)
        assert sharding.pack_share_into(dst, items, 24, base, 0) == len(ref)
        np.testing.assert_array_equal(dst[: len(ref)], ref)
        assert (dst[len(ref):] == 0xAB).all()


# ---- every rank ingests its own scenes: the only exchange is the gather of ids / votes (sharding.ResultGather) --------------------
def gather_worker(rank: int, port: int, outfile: str):
    sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    import torch.distributed as dist

    from similari_amd import sharding

    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        rows = [37, 12][rank]                       # the ranks' shares differ in size
        scenes = [[20, 17], [12]][rank]             # ... and in their number of scenes
        rg = sharding.ResultGather(capacity_rows=40, depth=3)
        seen = {}
        for step in range(7):                       # more steps than buffers: every buffer is reused, each reuse waits for its gather
            outs, base = [], 1000 * step + 100 * rank
            for k, n in enumerate(scenes):
                outs.append((np.arange(base + 10 * k, base + 10 * k + n, dtype=np.uint64), np.full(n, (step + rank + k) % 2, np.uint8)))
            rg.push(outs)
            if step in (2, 6):                      # the root may look at any finished step
                rg.drain()
                if rank == 0:
                    last = rg.last([37, 12])
                    seen[f"s{step}_ids0"], seen[f"s{step}_v0"] = last[0]
                    seen[f"s{step}_ids1"], seen[f"s{step}_v1"] = last[1]
        rg.drain()
        assert rg.steps == 7 and rows == sum(scenes)
        with pytest.raises(ValueError):
            rg.push([(np.zeros(41, np.uint64), np.zeros(41, np.uint8))])   # a share beyond the agreed capacity is refused locally
        if rank == 0:
            np.savez(outfile, **seen)
    finally:
        dist.destroy_process_group()


def test_result_gather_delivers_every_ranks_answers_to_the_root(tmp_path):
    """Local ingest (bench.py --gpus N --ingest local): no request travels; per step one asynchronous gather of ids | votes.  World 2 on
    gloo: shares of different sizes, more steps than buffers in flight, the root reads the step it drained."""
    import torch.multiprocessing as mp

    outfile = str(tmp_path / "gathered.npz")
    mp.spawn(gather_worker, args=(free_port(), outfile), nprocs=WORLD, join=True)
    got = dict(np.load(outfile))
    for step in (2, 6):
        for rank, scenes in ((0, [20, 17]), (1, [12])):
            base = 1000 * step + 100 * rank
            ids = np.concatenate([np.arange(base + 10 * k, base + 10 * k + n, dtype=np.uint64) for k, n in enumerate(scenes)])
            votes = np.concatenate([np.full(n, (step + rank + k) % 2, np.uint8) for k, n in enumerate(scenes)])
            np.testing.assert_array_equal(got[f"s{step}_ids{rank}"], ids)
            np.testing.assert_array_equal(got[f"s{step}_v{rank}"], votes)


def test_cpu_share_deals_whole_cores_out_socket_by_socket():
    """sharding.cpu_share: two sockets x four cores x two hardware threads (threads c and c + 8 share a core), four ranks: every rank gets
    two whole cores of one socket, no CPU twice, nothing left over; CPUs outside the process's allowed set are never handed out."""
    from similari_amd import sharding

    def sib(c):
        core = c % 8
        return (0 if core < 4 else 1), [core, core + 8]

    shares = [sharding.cpu_share(r, 4, allowed=list(range(16)), siblings_of=sib) for r in range(4)]
    assert shares == [[0, 1, 8, 9], [2, 3, 10, 11], [4, 5, 12, 13], [6, 7, 14, 15]]
    allowed = [1, 2, 3, 5, 9, 10, 13]                      # a cpuset that cut cores in half
    shares = [sharding.cpu_share(r, 2, allowed=allowed, siblings_of=sib) for r in range(2)]
    assert sorted(c for s in shares for c in s) == allowed and not set(shares[0]) & set(shares[1])
    assert shares == [[1, 2, 9, 10], [3, 5, 13]]                  # four cores left, two each, hardware threads kept together
    assert sharding.cpu_share(0, 1, allowed=allowed, siblings_of=sib) == []          # one rank: left alone
    assert sharding.cpu_share(0, 8, allowed=[0, 8], siblings_of=sib) == []           # fewer cores than ranks: left alone
