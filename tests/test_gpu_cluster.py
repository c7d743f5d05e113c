"""Scene sharding on the GPU box: the in-process dispatcher (sa_cluster_*: one engine + one worker thread per shard, scenes routed
scene_id % n) and the torch.distributed layers of similari_amd.sharding driving the HIP engine / the HIP BatchSort (world 1 here;
the world-2 run is the gloo test on CPU).  One GPU is enough for both: a cluster may place several shards on one device."""
import os
import socket

import numpy as np
import pytest

import oracle_lib as O
from similari_amd import abi, sharding, synth
from similari_amd import trackers as TR
from similari_amd.engine import Cluster, Engine, EngineError

pytestmark = pytest.mark.gpu


def scenes_and_cfg(seed, visual):
    rng = np.random.default_rng(seed)
    d = 64
    sizes = ((90, 100), (40, 70), (0, 30), (130, 120), (65, 64), (33, 90), (77, 50))
    ids = (4, 9, 10, 15, 21, 22, 31)
    if visual:
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual="cosine", visual_threshold=0.2, feature_len=d, max_observations=2,
                              visual_min_votes=1, visual_minimal_track_length=1, positional_min_confidence=0.1, max_idle_epochs=5)
        scs = {s: synth.visual_scene(rng, t, n, d, 2, canvas=(1000.0, 800.0), new_fraction=0.1) for s, (n, t) in zip(ids, sizes)}
    else:
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
        scs = {s: synth.sort_scene(rng, t, n, canvas=(1000.0, 800.0), oriented=(s % 2 == 0)) for s, (n, t) in zip(ids, sizes)}
    return cfg, scs


def tracks_of(sc, visual):
    kw = dict(feats=sc["track_feats"], feat_present=sc["track_present"]) if visual else {}
    return abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"], **kw)


def dets_of(sc, visual):
    kw = dict(feats=sc["det_feats"], feat_quality=sc["det_quality"]) if visual else {}
    return abi.make_detections(sc["det_boxes"], **kw)


@pytest.mark.parametrize("visual", [False, True], ids=["batch_sort", "batch_visual_sort"])
@pytest.mark.parametrize("shards", [1, 3])
def test_cluster_routes_scenes_and_matches_the_oracle(visual, shards):
    cfg, scs = scenes_and_cfg(81 + shards, visual)
    cl = Cluster(cfg, devices=[0] * shards)
    try:
        assert len(cl) == shards and cl.shard_of(22) == 22 % shards
        trs = {s: tracks_of(sc, visual) for s, sc in scs.items()}
        for s in scs:
            cl.upsert(s, trs[s])
        dets = {s: dets_of(sc, visual) for s, sc in scs.items()}
        for order in (list(scs), list(scs)[::-1]):  # res[i] belongs to req[i] whatever the routing does to the order
            req, res, outs = Engine.make_requests([(s, 1, dets[s]) for s in order])
            cl.associate_batch(req, res)
            for s, (ids, votes) in zip(order, outs):
                ref = O.associate(cfg, trs[s], 1, dets[s], want_matrices=False)
                np.testing.assert_array_equal(ids, ref["track_id"], err_msg=f"scene {s}")
                np.testing.assert_array_equal(votes, ref["voting_type"])
        ms = cl.last_ms()
        busy = {s % shards for s in scs}
        assert len(ms) == shards and all((m > 0.0) == (k in busy) for k, m in enumerate(ms)), ms
        # every scene's table lives on exactly one shard
        for s in scs:
            counts = [cl.engine(k).count(s) for k in range(shards)]
            assert counts[s % shards] == len(scs[s]["track_ids"]) and sum(counts) == counts[s % shards]
        # removing tracks goes to the owner too
        victim = 15
        cl.remove(victim, scs[victim]["track_ids"][:5])
        assert cl.engine(victim % shards).count(victim) == len(scs[victim]["track_ids"]) - 5
        # a failing shard reports which one and why; the others still completed
        bad = dets_of(scs[4], visual)
        bad.boxes[0].height = -1.0
        req, res, _ = Engine.make_requests([(9, 1, dets[9]), (4, 1, bad)])
        with pytest.raises(EngineError) as ei:
            cl.associate_batch(req, res)
        assert ei.value.code == abi.SA_ERR_BAD_ARG and "height" in str(ei.value)
    finally:
        cl.close()


def _init_single_rank_group():
    import torch.distributed as dist

    if not dist.is_initialized():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=0, world_size=1)
    return dist


def test_sharded_associator_over_the_hip_engine():
    """similari_amd.sharding.ShardedAssociator (array-level scatter / gather) with the HIP engine as the rank's engine, world 1."""
    _init_single_rank_group()
    cfg, scs = scenes_and_cfg(91, True)
    eng = Engine(cfg)
    try:
        sh = sharding.ShardedAssociator(eng, capacity_bytes=1 << 20, capacity_rows=1024)
        for s, sc in scs.items():
            sh.upsert_arrays(s, ids=sc["track_ids"], boxes=sc["track_boxes"], epochs=sc["track_epochs"], feats=sc["track_feats"],
                             feat_present=sc["track_present"])
        items = [(s, 1, sc["det_boxes"], sc["det_feats"], sc["det_quality"]) for s, sc in scs.items()]
        for _ in range(2):
            res = sh.associate(items)
            for (s, sc), (ids, votes) in zip(scs.items(), res):
                ref = O.associate(cfg, tracks_of(sc, True), 1, dets_of(sc, True), want_matrices=False)
                np.testing.assert_array_equal(ids, ref["track_id"], err_msg=f"scene {s}")
                np.testing.assert_array_equal(votes, ref["voting_type"])
        assert sh.last_local_ms > 0.0
    finally:
        eng.close()


def test_sharded_batch_tracker_over_the_hip_batch_sort():
    """similari_amd.sharding.ShardedBatchTracker with the HIP BatchSort as the local tracker (world 1): frame by frame the tracks of
    the oracle's BatchSort (ids included: global id == local id at world 1)."""
    _init_single_rank_group()
    from golden import make_golden as G

    rng = np.random.default_rng(92)
    hip = TR.BatchSort(bbox_history=3, max_idle_epochs=3, method=TR.PositionalMetricType.iou(0.3), device=0)
    o, keep = TR.sort_options(3, 3, TR.PositionalMetricType.iou(0.3), 0.05, None, 1.0 / 20.0, 1.0 / 160.0, batch=True)
    ora = O.OracleTracker(o, keep)
    try:
        sh = sharding.ShardedBatchTracker(hip, feature_len=0)
        world = {s: synth.dense_boxes(rng, 25, (700.0, 500.0), oriented=(s == 6)) for s in (2, 6, 11)}
        for f in range(5):
            batch = TR.PredictionBatchRequest()
            for s in world:
                world[s] = synth.jitter_boxes(rng, world[s], 1.5, angle_sigma=0.01 if s == 6 else 0.0)
                for b in world[s][rng.permutation(25)[: 25 - f]]:
                    batch.add(s, (TR.Universal2DBox(float(b["xc"]), float(b["yc"]), float(b["angle"]) if b["has_angle"] else None, float(b["aspect"]),
                                                    float(b["height"]), float(b["confidence"])), None))
            got, want = sh.predict(batch), ora.predict_batch(batch)
            assert list(got) == list(want)
            for s in got:
                np.testing.assert_array_equal(np.nan_to_num(G.track_rows(got[s]), nan=-7.0), np.nan_to_num(G.track_rows(want[s]), nan=-7.0),
                                              err_msg=f"frame {f} scene {s}")
    finally:
        hip.close()
        ora.close()


def test_result_gather_over_rccl_in_a_group_of_one():
    """similari_amd.sharding.ResultGather on its GPU path — pinned staging rows, the copy to device rows on torch's stream, an asynchronous
    `dist.gather` over RCCL, rows reused `depth` pushes later — in an RCCL group of this one rank (loopback=True makes the gather happen);
    more pushes than rows in flight, every step's payload distinct, the last one read back at the root."""
    import torch

    dist = _init_single_rank_group()
    torch.cuda.set_device(0)
    g = dist.new_group(ranks=[0], backend="nccl")
    try:
        rg = sharding.ResultGather(300, group=g, loopback=True, depth=3)
        assert rg.device.type == "cuda" and rg.copied is not None
        rng = np.random.default_rng(93)
        sent = None
        for step in range(11):
            outs = [(rng.integers(0, 2**63, n, dtype=np.uint64), rng.integers(0, 2, n).astype(np.uint8)) for n in (120, 0, 77, 100)]
            rg.push(outs)
            sent = outs
        rg.drain()
        (ids, votes), = rg.last([297])
        np.testing.assert_array_equal(ids, np.concatenate([o[0] for o in sent]))
        np.testing.assert_array_equal(votes, np.concatenate([o[1] for o in sent]))
        with pytest.raises(ValueError):
            rg.push([(np.zeros(301, np.uint64), np.zeros(301, np.uint8))])
    finally:
        dist.destroy_process_group(g)


@pytest.mark.parametrize("workload", ["c3", "c2"])
def test_bench_under_torchrun_takes_the_rccl_branch(workload):
    """bench.py the way the driver launches it for N > 1 (python -m torch.distributed.run ... bench.py --gpus N), here with ONE rank
    and SA_BENCH_FORCE_DIST=1 so that the process group is RCCL ("nccl") on this one-GPU box: communicator set-up, the barrier /
    all-reduce of the timed region, and the dispatch pass — ShardedAssociator's scatter -> one sa_associate_batch per rank (feature rows
    read in place from the registered receive buffer) -> gather — must run and give the resident run's answers."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "1", "--workload", workload, "--steps", "5", "--warmup", "2", "--profile-iters", "3",
           "--no-cpu-baseline", "--no-oracle", "--no-h2d"]
    r = subprocess.run(cmd, env=dict(os.environ, SA_BENCH_FORCE_DIST="1"), capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0
    disp = d["dispatch"]
    assert disp["backend"] == "nccl" and disp["ranks"] == 1
    assert disp["rank0_answers_match_resident_run"] is True


@pytest.mark.parametrize("workload,scenes", [("c3", 6), ("c2", 4)])
def test_bench_fixed_scene_set_over_two_ranks_on_one_device(workload, scenes):
    """bench.py --gpus 2 as the driver launches it, rehearsed on this one-GPU box (SA_BENCH_ONE_DEVICE=1: both ranks drive device 0,
    collectives over gloo): a FIXED set of scenes split scene_id % 2 (the default workload in its batched form c2b), `value` = total
    cells / wall time of every rank ingesting and running ITS scenes + the gather of ids / votes ("scaling": "strong"), value_scatter = rank
    0's scatter + per-rank batch + gather, the per-rank replay as value_resident; every scene's answer matches the synthetic truth, every
    rank's local-ingest answers match its resident run and reach the root."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--workload", workload, "--scenes", str(scenes), "--steps", "4", "--warmup", "2",
           "--profile-iters", "2", "--no-cpu-baseline", "--no-oracle", "--no-h2d"]
    r = subprocess.run(cmd, env=dict(os.environ, SA_BENCH_ONE_DEVICE="1"), capture_output=True, text=True, cwd=root, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 4
    assert d["config"]["scenes_total"] == scenes and d["config"]["scenes_per_gpu"] == scenes // 2
    disp = d["dispatch"]
    assert disp["ranks"] == 2 and disp["scenes"] == scenes and disp["steps"] == 4
    assert disp["rank0_answers_match_resident_run"] is True
    assert disp["match_accuracy_all_scenes"] > 0.9
    # `value`: every rank ingests the detections of ITS scenes, only ids / votes are gathered; the rank-0 scatter stays beside it
    loc = d["ingest_local"]
    lead = loc.get("device_features") or loc["host_boxes"]
    assert abs(d["value"] - lead["pairs_per_s"]) <= 1e-6 * d["value"] and d["value"] > 0
    assert abs(d["value_scatter"] - disp["pairs_per_s"]) <= 1e-6 * d["value_scatter"]
    for k in ("device_features", "host_pinned", "host_boxes"):
        if k in loc:
            assert loc[k]["every_rank_matches_its_resident_run"] is True and loc[k]["root_received_every_ranks_answers"] is True, (k, loc[k])
            assert loc[k]["steps"] >= 4 and loc[k]["gathered_bytes_per_step"] > 0
    assert d["value_resident"] > d["value_scatter"] > 0
    if workload == "c2":   # BASELINE's multi-GPU configuration rides beside the headline one
        c3 = d["c3_batchsort"]["host_boxes"]
        assert c3["every_rank_matches_its_resident_run"] is True and c3["root_received_every_ranks_answers"] is True and c3["pairs_per_s"] > 0
