#!/usr/bin/env python
"""bench.py — assoc-pairs/sec of the association hot path on MI355X (BASELINE.json metric).

A step = one pass of the hot path (pair pre-filter + cost matrices + quantise + assignment) over one request set.
  N = 1 (default): BASELINE C2, one scene-frame whose inputs are already resident in HBM (value = value_resident; value_h2d = the
      same frame ingested from host buffers every step).
  N > 1: scenes are independent (compatible() is false across scene ids, sort.rs:251), so the path shards by scene.  The default
      workload becomes its batched form: a FIXED set of 64 scenes of the C2 frame split scene_id % N (strong scaling).  `value` = local
      ingest: every rank owns the detections of ITS scenes — a step = each rank stages its boxes (feature rows resident in HBM), runs
      its share, and the ids / vote types are gathered on rank 0 (RCCL), total cells / wall time, maximum over ranks.  Beside it:
      value_h2d (feature rows in pinned host blocks), value_scatter (rank 0 packs and scatters the whole request set: one ingest
      point), value_resident (the per-rank replay), and BASELINE's multi-GPU configuration as `c3_batchsort`.  Workloads that are one
      frame (c4, c5, ...) are replicated per rank instead ("weak", no data-path collective) and carry the scatter / gather pass as the
      side object `dispatch`.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c2b|c2t|c3|c4|c5|...] [--scenes S] [--ingest local|scatter|both]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from similari_amd import abi, synth  # noqa: E402

DEFAULT_FLAGS = 0            # engine flags of the timed pass
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 dense peak
VALU_F32_PEAK_TFLOPS = 157.3  # f32 vector peak (256 CUs x 128 lanes x 2 flop x 2.4 GHz)
VALU_F64_PEAK_TFLOPS = 78.6  # f64 vector peak (MI355X_MICROARCH.md)


# Workloads that are a SET of independent scenes (BatchSort / BatchVisualSort request sets): under --gpus N > 1 the set is FIXED
# (--scenes, default 64: BASELINE C3's count) and split scene_id % N over the ranks — strong scaling — and the default workload c2
# becomes its batched form c2b (64 scenes of the C2 frame).  Scene s is generated from seed 1000 * seed + s on whichever rank owns it.
SCENE_SETS = {"c2b": 8, "c2bk3": 8, "c3": 8}   # scenes per request set at one GPU


def scene_ids_of_rank(total: int, world: int, rank: int):
    """The scenes of a fixed set that rank `rank` of `world` owns: scene_id % world (similari_amd.sharding.owner, sa_cluster_shard_of)."""
    return [s for s in range(int(total)) if s % int(world) == int(rank)]


def workload(name: str, seed: int, scene_ids=None):
    """Returns (config, scene dicts, description).  C2 is the configuration the metric is quoted on.  scene_ids (scene-set workloads):
    generate exactly these scenes of the set."""
    rng = np.random.default_rng(seed)
    # (the default set of a one-GPU run keeps the scenes of earlier rounds: ONE generator streamed over the scenes; an explicit list of
    # scene ids — the fixed set of a multi-GPU run — draws scene s from its own seed, so that any rank can generate exactly its scenes)
    legacy = name in SCENE_SETS and scene_ids is None
    if legacy:
        scene_ids = list(range(SCENE_SETS[name]))
    srng = (lambda sid: rng) if legacy else (lambda sid: np.random.default_rng(1000 * seed + sid))
    if name == "c2":
        n = t = 1000
        d, k = 512, 1
        sc = synth.visual_scene(rng, t, n, d, k)
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual="cosine", visual_threshold=0.2, feature_len=d,
                              max_observations=k, visual_min_votes=1, visual_minimal_track_length=1,
                              positional_min_confidence=0.1, max_idle_epochs=5)
        return cfg, [sc], "VisualSORT 1000 tracks x 1000 dets, 512-d cosine + IoU(0.3), K=1 (BASELINE C2)"
    if name in ("c2b", "c2bk3"):
        # BatchVisualSORT (visual_sort/batch_api.rs:213-317): S scenes of the C2 frame in ONE request set (grid.z = scene)
        n = t = 1000
        d, k = 512, (3 if name.endswith("k3") else 1)
        scs = [synth.visual_scene(srng(sid), t, n, d, k) for sid in scene_ids]
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual="cosine", visual_threshold=0.2, feature_len=d,
                              max_observations=k, visual_min_votes=1, visual_minimal_track_length=1,
                              positional_min_confidence=0.1, max_idle_epochs=5)
        return cfg, scs, f"BatchVisualSORT scenes of (1000 tracks x 1000 dets, 512-d cosine + IoU(0.3), K={k}) in one request set per GPU (BASELINE C2, batched: visual_sort/batch_api.rs)"
    if name == "c2t":
        # the frame a VisualSORT tracker loop hands over once idle tracks linger (max_idle_epochs): more table rows than detections
        n, t = 1000, 1500
        d, k = 512, 1
        sc = synth.visual_scene(rng, t, n, d, k)
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual="cosine", visual_threshold=0.2, feature_len=d,
                              max_observations=k, visual_min_votes=1, visual_minimal_track_length=1,
                              positional_min_confidence=0.1, max_idle_epochs=5)
        return cfg, [sc], "VisualSORT 1000 dets x 1500 tracks (a tracker loop's table with idle tracks), 512-d cosine + IoU(0.3), K=1"
    if name in ("c2k3", "c2k2"):
        n = t = 1000
        d, k = 512, int(name[-1])
        sc = synth.visual_scene(rng, t, n, d, k)
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual="cosine", visual_threshold=0.2, feature_len=d,
                              max_observations=k, visual_min_votes=1, visual_minimal_track_length=1,
                              positional_min_confidence=0.1, max_idle_epochs=5)
        return cfg, [sc], "VisualSORT 1000 tracks x 1000 dets, 512-d cosine + IoU(0.3), K=" + str(k) + " observations per track (BASELINE C2, deeper bank)"
    if name == "c2e":
        n = t = 1000
        d, k = 512, 1
        sc = synth.visual_scene(rng, t, n, d, k)
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual="euclidean", visual_threshold=0.5, feature_len=d,
                              max_observations=k, visual_min_votes=1, visual_minimal_track_length=1,
                              positional_min_confidence=0.1, max_idle_epochs=5)
        return cfg, [sc], "VisualSORT 1000 tracks x 1000 dets, 512-d EUCLIDEAN + IoU(0.3), K=1 (the C2 frame with the other visual metric)"
    if name in ("c2d", "c2dd"):
        # the C2 frame under the reference's DEFAULT VisualSortOptions (visual_sort/options.rs:194-205, metric/builder.rs:26-42):
        # euclidean metric, five observations per track, visual_minimal_track_length 3 (the threshold stays finite so that is_ok prunes)
        n = t = 1000
        d, k = 512, 5
        sc = synth.visual_scene(rng, t, n, d, k)
        # c2dd: the literal default threshold as well (f32::MAX: every cell present, every pair a group of the vote)
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual="euclidean", visual_threshold=0.5 if name == "c2d" else 3.4028234663852886e38,
                              feature_len=d, max_observations=k, visual_min_votes=1, visual_minimal_track_length=3,
                              positional_min_confidence=0.1, max_idle_epochs=2)
        return cfg, [sc], "VisualSORT 1000 x 1000 x 512-d under the reference's default options: EUCLIDEAN, 5 observations per track, minimal track length 3"
    if name == "c5":
        t, n, d, k = 5000, 2000, 4096, 1
        sc = synth.visual_scene(rng, t, n, d, k, canvas=(7680.0, 4320.0))
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual="cosine", visual_threshold=0.2, feature_len=d,
                              max_observations=k, visual_min_votes=1, visual_minimal_track_length=1,
                              positional_min_confidence=0.1, max_idle_epochs=5)
        return cfg, [sc], "VisualSORT 5000 tracks x 2000 dets, 4096-d cosine + IoU(0.3) (BASELINE C5)"
    if name == "c4":
        sc = synth.sort_scene(rng, 2000, 2000, canvas=(8192.0, 8192.0), oriented=True)
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
        return cfg, [sc], "Oriented SORT 2000 x 2000, rotated IoU clipping (BASELINE C4)"
    if name == "c3":
        scs = [synth.sort_scene(srng(sid), 500, 500, canvas=(4096.0, 4096.0)) for sid in scene_ids]
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
        return cfg, scs, "BatchSORT IoU, scenes of 500 x 500 in one request set per GPU (BASELINE C3: 64 scenes over 8 GPUs)"
    if name == "c1":
        sc = synth.sort_scene(rng, 100, 100, canvas=(1920.0, 1080.0))
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
        return cfg, [sc], "SORT IoU 100 x 100, dense variant (BASELINE C1)"
    if name in ("c1ref", "c1ref100"):
        # the reference's OWN bench layout (benches/simple_sort_iou_tracker.rs:29-61: object i at (1000 i, 1000 i), 50 x 50, drift 1 px,
        # spatio-temporal constraint (1, 1.0), IoU(0.3)), 500 (or 100) objects: nearly every off-diagonal pair dies in the pre-filter, so
        # "pairs/s" here is the NOMINAL objects^2 / time that sits beside assets/benchmarks/benchmarks.md:36-40 (13.4 M nominal pairs/s
        # for the whole predict() at 500 objects on four laptop cores) — the association alone on this side
        n = 100 if name.endswith("100") else 500
        tb = synth.diagonal_boxes(n)
        db = synth.jitter_boxes(rng, tb, pos_sigma=1.0, size_rel=0.001)
        perm = rng.permutation(n)
        sc = dict(track_ids=np.arange(1, n + 1, dtype=np.uint64), track_boxes=tb, track_epochs=np.zeros(n, np.uint64), det_boxes=db[perm],
                  truth=(perm + 1).astype(np.uint64))
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=1, constraints=[(1, 1.0)])
        return cfg, [sc], f"SORT IoU {n} x {n} in the reference bench layout (objects 1000 px apart, constraint (1, 1.0)): benches/simple_sort_iou_tracker.rs"
    if name in ("c1ref_or", "vref"):
        # the reference's other bench layouts (the whole predict() in these scenarios: scripts/bench_reference_layouts.py, beside
        # assets/benchmarks/benchmarks.md:42-131): c1ref_or = benches/simple_sort_iou_tracker_oriented.rs (500 objects, a random angle in [0, 1)
        # per box); vref = benches/simple_visual_sort_tracker.rs:98-141 (100 objects 20 x 50, features 10 i + U(-0.01, 0.01), Euclidean(10.0),
        # three observations per track, min votes 2, constraint (1, 1.0)) — the association of one such frame
        n = 500 if name == "c1ref_or" else 100
        tb = synth.diagonal_boxes(n, w=50.0 if name == "c1ref_or" else 20.0)
        if name == "c1ref_or":
            tb["angle"] = rng.uniform(0.0, 1.0, n).astype(np.float32); tb["has_angle"] = 1
        db = synth.jitter_boxes(rng, tb, pos_sigma=1.0, size_rel=0.001)
        if name == "c1ref_or":
            db["angle"] = rng.uniform(0.0, 1.0, n).astype(np.float32)
        perm = rng.permutation(n)
        sc = dict(track_ids=np.arange(1, n + 1, dtype=np.uint64), track_boxes=tb, track_epochs=np.zeros(n, np.uint64), det_boxes=db[perm],
                  truth=(perm + 1).astype(np.uint64))
        if name == "c1ref_or":
            cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=1, constraints=[(1, 1.0)])
            return cfg, [sc], "Oriented SORT IoU 500 x 500 in the reference bench layout (benches/simple_sort_iou_tracker_oriented.rs)"
        d, k = 512, 3
        base = 10.0 * np.arange(n, dtype=np.float32)[:, None]
        sc["track_feats"] = (base[:, None, :] + rng.uniform(-0.01, 0.01, (n, k, d))).astype(np.float32)
        sc["track_present"] = np.ones((n, k), np.uint8)
        sc["det_feats"] = (base + rng.uniform(-0.01, 0.01, (n, d))).astype(np.float32)[perm]
        sc["det_quality"] = np.ones(n, np.float32)
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual="euclidean", visual_threshold=10.0, feature_len=d,
                              max_observations=k, visual_min_votes=2, visual_minimal_track_length=1, positional_min_confidence=0.1,
                              max_idle_epochs=1, constraints=[(1, 1.0)], visual_minimal_own_area_percentage_use=0.5,
                              visual_minimal_own_area_percentage_collect=0.6)
        return cfg, [sc], "VisualSORT 100 x 100 x 512-d in the reference bench layout: Euclidean(10.0), 3 observations, min votes 2 (benches/simple_visual_sort_tracker.rs)"
    if name == "sd":
        # not a BASELINE config: a crowd for plain SORT (the C2 frame without features) — the positional vote alone has to untangle
        # the overlaps, so its graph has large connected components
        sc = synth.sort_scene(rng, 1000, 1000, canvas=(1920.0, 1080.0))
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
        return cfg, [sc], "SORT IoU 1000 x 1000 on the C2 canvas (crowd: large components in the positional vote)"
    if name == "sdt":
        # not a BASELINE config: the frame a SORT tracker loop hands over in a crowd — 1000 detections against a table of 2500 tracks
        # (live and idle ones): beyond the one-workgroup tail, dozens of components of 9..32 rows (the general tail's middle tier)
        sc = synth.sort_scene(rng, 2500, 1000, canvas=(1920.0, 1080.0))
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
        return cfg, [sc], "SORT IoU 1000 detections x 2500 tracks on the C2 canvas (a tracker loop's crowd frame: general tail, mid-sized components)"
    if name == "c2n":
        # C2 with 15 % new objects and 10 % of the detections below the quality gate: the positional (Hungarian) stage has real
        # work after the visual vote inside the timed region
        n = t = 1000
        d, k = 512, 1
        sc = synth.visual_scene(rng, t, n, d, k, new_fraction=0.15)
        sc["det_quality"][rng.uniform(size=n) < 0.10] = 0.05
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual="cosine", visual_threshold=0.2, feature_len=d,
                              max_observations=k, visual_min_votes=1, visual_minimal_track_length=1, visual_minimal_quality_use=0.3,
                              positional_min_confidence=0.1, max_idle_epochs=5)
        return cfg, [sc], "VisualSORT 1000 x 1000 x 512-d cosine + IoU(0.3), 15 % new objects, 10 % featureless detections (non-empty Hungarian stage)"
    if name == "giant":
        # not a BASELINE config: ONE connected component — 640 boxes piled on each other, IoU threshold 0.05 — the case a CPU
        # kuhn_munkres handles at its own speed whatever the density (sort/voting.rs:86)
        sc = synth.sort_scene(rng, 640, 640, canvas=(150.0, 150.0), pos_sigma=10.0)
        cfg = abi.make_config(positional="iou", positional_threshold=0.05, max_idle_epochs=5)
        return cfg, [sc], "SORT IoU(0.05) 640 x 640, all boxes on one pile, 10 px of jitter: one connected component of the positional vote, ~140 k usable edges, a third of the rows lose their greedy bid"
    if name == "bigcrowd":
        # beyond the one-workgroup tail: 1500 detections in a crowd, 12 px of jitter (345 rows lose their greedy bid; components of
        # tens to hundreds of rows go to the general tail's wave-cooperative solver, state in HBM)
        sc = synth.sort_scene(rng, 1500, 1500, canvas=(1000.0, 800.0), pos_sigma=12.0)
        cfg = abi.make_config(positional="iou", positional_threshold=0.15, max_idle_epochs=5)
        return cfg, [sc], "SORT IoU(0.15) 1500 x 1500 in a crowd, 12 px of jitter: the general assignment tail with big components"
    if name == "bigpile":
        sc = synth.sort_scene(rng, 1200, 1200, canvas=(200.0, 200.0), pos_sigma=10.0)
        cfg = abi.make_config(positional="iou", positional_threshold=0.05, max_idle_epochs=5)
        return cfg, [sc], "SORT IoU(0.05) 1200 x 1200 on one pile: ONE component of 1200 rows and ~340 k edges in the general assignment tail"
    if name == "c3m":
        # BASELINE C3's Mahalanobis half: the Kalman states come from the product's own device-side upkeep (three frames through
        # the BatchSort facade), see maha_engine() below — no synthetic scene dict
        return None, None, "BatchSORT Mahalanobis, 8 scenes x 500 x 500 per GPU (BASELINE C3: 64 scenes over 8 GPUs)"
    raise SystemExit(f"unknown workload {name}")


def maha_engine(local_rank, seed, n_scenes=8, n=500):
    """A BatchSort(Mahalanobis) facade with device-side upkeep, three frames in: its engine then holds n tracks per scene with
    genuine Kalman states and a fourth frame staged.  Returns (facade, borrowed Engine, cells per step)."""
    from similari_amd import trackers as TR
    from similari_amd.engine import Engine

    rng = np.random.default_rng(seed)
    trk = TR.BatchSort(bbox_history=2, max_idle_epochs=5, method=TR.PositionalMetricType.maha(), device=local_rank, device_upkeep=True)
    world = {s: synth.dense_boxes(rng, n, (4096.0, 4096.0)) for s in range(n_scenes)}
    for f in range(4):
        req = TR.PredictionBatchRequest()
        for s in range(n_scenes):
            world[s] = synth.jitter_boxes(rng, world[s], 1.5)
            for b in world[s]:
                req.add(s, (TR.Universal2DBox(float(b["xc"]), float(b["yc"]), None, float(b["aspect"]), float(b["height"]), float(b["confidence"])), None))
        res = trk.predict(req)
    cont = sum(1 for s in range(n_scenes) for t in res[s] if t.length > 1)
    eng = Engine.borrowed(trk.lib, trk.lib.sa_tracker_engine(trk.h))
    return trk, eng, n_scenes * n * n, cont / float(n_scenes * n)


def stage(eng, cfg, scenes, keys=None):
    """Upserts every scene's tracks and stages its detections as one request set; keys = the scenes' ids (default 0, 1, ...)."""
    visual = cfg.visual_kind != abi.SA_VIS_NONE
    keep = []
    keys = list(range(len(scenes))) if keys is None else list(keys)
    for s, sc in zip(keys, scenes):
        kw = dict(feats=sc["track_feats"], feat_present=sc["track_present"]) if visual else {}
        tr = abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"], **kw)
        eng.upsert(s, tr)
        keep.append(tr)
    eng.batch_begin()
    dets = []
    for s, sc in zip(keys, scenes):
        kw = dict(feats=sc["det_feats"], feat_quality=sc["det_quality"]) if visual else {}
        d = abi.make_detections(sc["det_boxes"], **kw)
        eng.batch_add(s, 1, d)
        dets.append(d)
    return keep, dets


def pmc_traffic(workload: str, kernel: str):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r01_*_pmc_traffic.json, written by
    scripts/pmc_traffic.sh on the GPU box: separate FETCH_SIZE / WRITE_SIZE passes, FETCH doubled as the gfx950 guide says).
    Counters cannot be read from inside this process; None when no pass exists for this workload."""
    best = None
    files = sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"))
    newest_round = files[-1].name.split("_")[0] if files else None
    for f in files:
        try:
            d = json.loads(f.read_text())["workloads"].get(workload, {}).get(kernel)
        except Exception:
            d = None
        if d:
            # (a pass of an EARLIER round than the newest committed one says nothing about this tree's launch — tile orders and kernels
            # change between rounds: it is reported, but labelled as stale)
            best = (d["hbm_bytes"], f.name, f.name.split("_")[0] != newest_round)
    return best


def launch_floor_us():
    """What ONE dependent launch costs on this stack whatever it computes (scripts/micro/launch_floor.hip: an empty 1000-block kernel in a
    chain of 200 launches on one stream), from the newest committed measurement profiles/r*_launch_floor.txt; None when there is none."""
    import re

    files = sorted((ROOT / "profiles").glob("r*_launch_floor.txt"))
    if not files:
        return None
    for ln in files[-1].read_text().splitlines():
        m = re.match(r"^empty\s+blocks\s+1000:\s+([0-9.]+) us per launch in a chain", ln)
        if m:
            return float(m.group(1)), files[-1].name
    return None


def rocprof_stats(workload: str):
    """Average kernel durations (us) of the rocprofv3 --kernel-trace --stats summary committed for this workload
    (profiles/r*_<workload>_kernel_stats.csv, the newest round wins): {engine kernel name: avg us}."""
    import csv

    out = {}
    files = sorted((ROOT / "profiles").glob(f"r*_{workload}_kernel_stats.csv"))
    if not files:
        return out
    names = ("k_frame_visual", "k_frame", "k_visual_cost", "k_visual_cosine", "k_visual_euclid", "k_bestfit_tile", "k_bestfit_resolve", "k_assign_small",
             "k_assign_label", "k_assign_solve")
    try:
        for row in csv.DictReader(files[-1].open()):
            nm = row.get("Name", "")
            for k in names:
                if nm.startswith("void " + k + "<") or nm.startswith("void " + k + "(") or nm.startswith(k + "<") or nm.startswith(k + "(") or (k == "k_assign_small" and "k_assign_small2<" in nm):
                    key = "k_visual_cost" if k in ("k_visual_cosine", "k_visual_euclid") else k
                    avg = float(row.get("AverageNs", 0.0)) / 1e3
                    out[key] = max(out.get(key, 0.0), avg)  # several specialisations: the one this workload runs dominates
                    break
    except Exception:
        return {}
    return out


def cpu_solve_only(cfg, scenes, budget_s=8.0):
    """What the assignment tail replaces, ALONE: the reference's pathfinding::kuhn_munkres (oracle: or_kuhn_munkres) on the very
    N x (T + N) i64 matrix SortVoting::winners builds for this frame (sort/voting.rs:44-86: quantised weights, the new-track threshold
    on the diagonal of the self columns) — one host thread, matrix already built.  So that a solve is compared with a solve."""
    import ctypes as C

    import oracle_lib as O

    sc = scenes[0]
    tracks = abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"])
    det = abi.make_detections(sc["det_boxes"])
    ref = O.associate(cfg, tracks, 1, det, shards=int(max(1, min(os.cpu_count() or 1, 64))))
    q = ref["quantised"]
    n, t = q.shape
    thr_q = int(O.lib().or_quantise(cfg.positional_threshold)) if cfg.positional_kind == abi.SA_POS_IOU else 1000000
    w = np.zeros((n, t + n), np.int64)
    w[:, :t] = q
    w[np.arange(n), t + np.arange(n)] = thr_q
    w = np.ascontiguousarray(w)
    total = C.c_int64()
    assign = np.zeros(n, np.uint32)
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 2 or (time.perf_counter() < t_end and len(times) < 30):
        t0 = time.perf_counter()
        rc = O.lib().or_kuhn_munkres(n, t + n, w.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(total), assign.ctypes.data_as(C.POINTER(C.c_uint32)))
        times.append(time.perf_counter() - t0)
        assert rc == 0
    return {"ms": 1e3 * min(times), "runs": len(times), "cores": 1, "rows": int(n), "cols": int(t + n), "edges_above_threshold": int((q > thr_q).sum()),
            "what": "or_kuhn_munkres (the oracle's restatement of pathfinding::kuhn_munkres) on the N x (T + N) matrix of SortVoting::winners, matrix already "
                    "built, best run; compare with the assignment kernels' avg_us (k_assign_small, or k_assign_label + k_assign_solve)"}


def cpu_baseline(cfg, scenes, budget_s=12.0):
    """The oracle (reference-faithful per-pair recompute, 1 thread) on a bounded sample of the same workload."""
    import oracle_lib as O

    sc = scenes[0]
    visual = cfg.visual_kind != abi.SA_VIS_NONE
    T = len(sc["track_boxes"])
    kw = dict(feats=sc["track_feats"], feat_present=sc["track_present"]) if visual else {}
    tracks = abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"], **kw)
    # first a small probe to size the sample to ~budget_s of CPU work
    n_all = len(sc["det_boxes"])
    probe = min(n_all, 16)

    def run(n):
        kw = dict(feats=sc["det_feats"][:n], feat_quality=sc["det_quality"][:n]) if visual else {}
        det = abi.make_detections(sc["det_boxes"][:n], **kw)
        t0 = time.perf_counter()
        O.associate(cfg, tracks, 1, det, want_matrices=False)
        return time.perf_counter() - t0

    tp = max(run(probe), 1e-6)
    n = int(min(n_all, max(probe, budget_s / tp * probe)))
    # whole frames of the sample until ~budget_s of CPU work is spent (at least 1, at most 20), best time kept
    one = run(n)
    reps = int(max(1, min(20, budget_s / max(one, 1e-6))))
    times = [one] + [run(n) for _ in range(reps - 1)]
    dt = min(times)
    return {
        "value": n * T / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
        "sample": f"oracle or_associate (oracle/oracle.cpp, g++ {ORACLE_FLAGS}), {n} of {n_all} detections x {T} tracks of scene 0, "
                  f"best of {len(times)} runs ({sum(times):.1f} s of CPU work, 1 thread; host has {os.cpu_count()} cores)",
    }


ORACLE_FLAGS = "-O3 -march=x86-64-v3 -ffp-contract=off (the reference's own target-cpu, .cargo/config.toml)"


def cpu_baseline_threads(cfg, scenes, budget_s=6.0):
    """The same oracle with its distance stage on `shards` host threads partitioned track_id % shards the way the reference's
    TrackStore shards it (store.rs:490-493) and ONE vote after the shards (sort/simple_api.rs:147-162) — or_associate_sharded.
    Reported beside cpu_baseline (BASELINE.md section 2 asks for both variants); it is not the judged baseline object."""
    import oracle_lib as O

    sc = scenes[0]
    visual = cfg.visual_kind != abi.SA_VIS_NONE
    T, N = len(sc["track_boxes"]), len(sc["det_boxes"])
    shards = int(max(1, min(os.cpu_count() or 1, 64, T)))
    kw = dict(feats=sc["track_feats"], feat_present=sc["track_present"]) if visual else {}
    tracks = abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"], **kw)
    kw = dict(feats=sc["det_feats"], feat_quality=sc["det_quality"]) if visual else {}
    det = abi.make_detections(sc["det_boxes"], **kw)
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 2 or (time.perf_counter() < t_end and len(times) < 50):
        t0 = time.perf_counter()
        O.associate(cfg, tracks, 1, det, want_matrices=False, shards=shards)
        times.append(time.perf_counter() - t0)
    return {
        "value": N * T / min(times), "unit": "pairs/s", "cores": shards, "kind": "port",
        "sample": f"oracle or_associate_sharded (g++ {ORACLE_FLAGS}): distances on {shards} threads, tracks of scene 0 partitioned id % {shards} "
                  f"(store.rs:490-493), one vote after the shards, all {N} detections x {T} tracks per frame, best of {len(times)} frames; "
                  f"host has {os.cpu_count()} cores",
    }


def oracle_answers(cfg, scenes):
    """The oracle's ids / vote types for every scene of the timed frame (distance stage sharded over the host's cores)."""
    import oracle_lib as O

    visual = cfg.visual_kind != abi.SA_VIS_NONE
    shards = int(max(1, min(os.cpu_count() or 1, 64)))
    out = []
    for sc in scenes:
        kw = dict(feats=sc["track_feats"], feat_present=sc["track_present"]) if visual else {}
        tracks = abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"], **kw)
        kw = dict(feats=sc["det_feats"], feat_quality=sc["det_quality"]) if visual else {}
        det = abi.make_detections(sc["det_boxes"], **kw)
        r = O.associate(cfg, tracks, 1, det, want_matrices=False, shards=shards)
        out.append((r["track_id"], r["voting_type"]))
    return out


def f64_clip_model(scenes, sample=1500, seed=0):
    """Algorithmic f64 work of the positional tiles: the pairs that survive too_far() (bbox.rs:452-474) go through
    Sutherland-Hodgman (clipping.rs:12-91) and the shoelace area.  Counts the f64 operations of a restatement of that algorithm on a
    random sample of surviving pairs of scene 0 (side tests 7 flop, crossing points 18 flop incl. the division, shoelace 5 flop per
    vertex) and scales to all surviving pairs of the frame.  Returns (surviving pairs of the frame, mean flop per surviving pair)."""
    import math

    rng = np.random.default_rng(seed)
    total_pairs, flops_per_pair = 0, []
    for si, sc in enumerate(scenes):
        c, t = sc["det_boxes"], sc["track_boxes"]
        def rad(b):
            hw = (b["aspect"] * b["height"] / np.float32(2)).astype(np.float32)
            hh = (b["height"] / np.float32(2)).astype(np.float32)
            return np.sqrt(hw * hw + hh * hh)
        rc, rt = rad(c), rad(t)
        dx = c["xc"][:, None] - t["xc"][None, :]
        dy = c["yc"][:, None] - t["yc"][None, :]
        near = dx * dx + dy * dy <= (rc[:, None] + rt[None, :]) ** 2
        total_pairs += int(near.sum())
        if si:
            continue
        ii, jj = np.nonzero(near)
        if not len(ii):
            continue
        pick = rng.choice(len(ii), min(sample, len(ii)), replace=False)

        def verts(b):
            a = float(b["angle"]) if b["has_angle"] else 0.0
            cs, sn = math.cos(a), math.sin(a)
            hw, hh = float(b["height"]) * float(b["aspect"]) / 2.0, float(b["height"]) / 2.0
            x, y = float(b["xc"]), float(b["yc"])
            r1 = (-hw * cs - hh * sn, -hw * sn + hh * cs)
            r2 = (hw * cs - hh * sn, hw * sn + hh * cs)
            return [(x + r1[0], y + r1[1]), (x + r2[0], y + r2[1]), (x - r1[0], y - r1[1]), (x - r2[0], y - r2[1])]

        for k in pick:
            poly, clip, fl = verts(c[ii[k]]), verts(t[jj[k]]), 0
            for e in range(4):
                cs_, ce_ = clip[e - 1], clip[e]
                out = []
                if not poly:
                    break
                inside = lambda p: (ce_[0] - cs_[0]) * (p[1] - cs_[1]) - (ce_[1] - cs_[1]) * (p[0] - cs_[0]) <= 0.0
                for v in range(len(poly)):
                    s_, e_ = poly[v - 1], poly[v]
                    fl += 7  # one side test per vertex (its predecessor's is reused)
                    ie, is_ = inside(e_), inside(s_)
                    if ie != is_:
                        fl += 18  # compute_intersection: 3 differences, 2 cross products, 1 determinant, 1 division, 2 x (2 mul + 1 sub + 1 mul)
                        dcx, dcy, dpx, dpy = s_[0] - e_[0], s_[1] - e_[1], cs_[0] - ce_[0], cs_[1] - ce_[1]
                        n1, n2 = s_[0] * e_[1] - s_[1] * e_[0], cs_[0] * ce_[1] - cs_[1] * ce_[0]
                        n3 = dcx * dpy - dcy * dpx
                        if n3 != 0.0:
                            out.append(((n1 * dpx - n2 * dcx) / n3, (n1 * dpy - n2 * dcy) / n3))
                    if ie:
                        out.append(e_)
                poly = out
            fl += 5 * len(poly) + 4  # shoelace with the shift, halving, IoU epilogue
            flops_per_pair.append(fl)
    return total_pairs, (float(np.mean(flops_per_pair)) if flops_per_pair else 0.0)


def kernel_bytes_model(cfg, scenes, matched_edges):
    """Bytes the positional launch has to move whatever its implementation: the raw detection records in (48 B/box + the optional
    per-detection arrays), the track geometry + polygon (16 + 64 B/track), the derived candidate arrays out (16 + 64 + 4 B), one
    16-byte edge record per surviving cell of the vote, and for VisualSORT the candidates' feature rows in (and out, when padded)."""
    n_ = sum(len(s["det_boxes"]) for s in scenes)
    t_ = sum(len(s["track_boxes"]) for s in scenes)
    b = 48.0 * n_ + 80.0 * t_ + 84.0 * n_ + 16.0 * matched_edges
    return b


def h2d_pipelined(eng, cfg, scenes, iters, feats_on_device=False, keys=None, on_results=None, region=None):
    """The reference's predict() ingests host buffers every frame (visual_sort/simple_api.rs:130-170): the same frame through
    sa_pipe_submit / sa_pipe_wait, two request sets in flight, features in a block from sa_host_alloc (DMA'd in place), results
    copied out — H2D and D2H inside the timed region.  feats_on_device: the feature rows lie in device memory already (a ReID model on
    the same GPU: sa_device_block_register) and are read in place; boxes and qualities still arrive from the host every frame."""
    visual = cfg.visual_kind != abi.SA_VIS_NONE
    blocks, items, dev_blocks = [], [], []
    for s, sc in zip(list(range(len(scenes))) if keys is None else keys, scenes):
        kw = {}
        if visual and feats_on_device:
            import torch

            t = torch.from_numpy(np.ascontiguousarray(sc["det_feats"], np.float32)).to(f"cuda:{max(cfg.device, 0)}")
            torch.cuda.synchronize()
            eng.register_device_block(t.data_ptr(), t.numel() * 4, max(cfg.device, 0))
            dev_blocks.append(t)
            kw = dict(feats_device_ptr=t.data_ptr(), feat_quality=sc["det_quality"])
        elif visual:
            b = eng.host_block(sc["det_feats"].shape)
            b[...] = sc["det_feats"]
            blocks.append(b)
            kw = dict(feats=b, feat_quality=sc["det_quality"])
        items.append((s, 1, abi.make_detections(sc["det_boxes"], **kw)))
    depth = int(os.environ.get("SA_PIPE_DEPTH", "3"))  # tickets outstanding (the engine holds three banks)
    sets = [eng.make_requests(items) for _ in range(depth)]
    lib, h = eng.lib, eng.h
    import ctypes as C

    tk = [C.c_uint64() for _ in range(depth)]
    ns = len(items)
    host = [0.0, 0.0]  # seconds of host time inside sa_pipe_submit / sa_pipe_wait

    def loop(k):
        pc = time.perf_counter
        rc = 0
        for i in range(k + depth - 1):
            a = pc()
            if i < k:
                rc |= lib.sa_pipe_submit(h, ns, sets[i % depth][0], C.byref(tk[i % depth]))
            b = pc()
            j = i - (depth - 1)
            if j >= 0:
                rc |= lib.sa_pipe_wait(h, tk[j % depth], sets[j % depth][1])
                if on_results is not None:   # (a multi-GPU run whose ranks ingest their own scenes: the step's ids / votes go to the root)
                    on_results(sets[j % depth][2])
            c = pc()
            host[0] += b - a
            host[1] += c - b
        assert rc == 0, eng.lib.sa_last_error(h)

    loop(10)
    host[0] = host[1] = 0.0
    if region is not None:   # (several ranks: barrier + synchronize on both sides, the maximum over the ranks)
        dt = region(lambda: loop(iters))
    else:
        t0 = time.perf_counter()
        loop(iters)
        dt = time.perf_counter() - t0
    # the synchronous form for comparison (stage -> DMA -> kernels -> results, nothing overlapped)
    req, res, outs = sets[0]
    for _ in range(5):
        eng.associate_batch(req, res)
    t1 = time.perf_counter()
    for _ in range(max(10, iters // 4)):
        eng.associate_batch(req, res)
    dts = (time.perf_counter() - t1) / max(10, iters // 4)
    ids = [o[0].copy() for o in sets[(iters - 1) % depth][2]]
    for b in blocks:
        eng.host_free(b)
    for t in dev_blocks:
        eng.unregister_device_block(t.data_ptr())
    cells = sum(len(s["det_boxes"]) * len(s["track_boxes"]) for s in scenes)
    h2d_bytes = sum(len(s["det_boxes"]) * (48 + ((0 if feats_on_device else 4 * cfg.feature_len) + 4 if visual else 0)) for s in scenes)
    return {"pairs_per_s": cells * iters / dt, "ms_per_step": 1e3 * dt / iters, "steps": iters,
            "h2d_bytes_per_step": h2d_bytes, "d2h_bytes_per_step": 9 * sum(len(s["det_boxes"]) for s in scenes),
            "synchronous_ms_per_step": 1e3 * dts,
            "tickets_in_flight": depth, "host_us_in_submit": 1e6 * host[0] / max(1, iters), "host_us_in_wait": 1e6 * host[1] / max(1, iters),
            "note": ("sa_pipe_submit / sa_pipe_wait, three request sets in flight, boxes and qualities from host buffers every frame, the feature rows read in place "
                     "from a registered device block (sa_device_block_register: what a ReID model on the same GPU leaves behind), results copied out"
                     if feats_on_device else
                     "sa_pipe_submit / sa_pipe_wait, three request sets in flight (H2D of frame n+1 beside the kernels of frame n, frame n+2 queued), features in a "
                     "sa_host_alloc block, results copied out; synchronous_ms_per_step = sa_associate_batch on the same buffers")}, ids


def local_ingest(wname, total_scenes, world, rank, local_rank, dist, barrier, max_over_ranks, iters):
    """--gpus N > 1, `--ingest local`: every rank OWNS the detections of its scenes (scene_id % N: a camera's detector feeds the GPU that owns
    the camera's scene) — nothing of the request travels between ranks; each step every rank ingests its own share (boxes from host buffers;
    the feature rows either resident in HBM — a ReID model on the same GPU: `device_features` — or in pinned host blocks: `host_pinned`),
    runs it, and the ids / vote types go to rank 0 (similari_amd.sharding.ResultGather: one KB-scale gather per step, asynchronous).
    EXACTLY `iters` steps between barrier + synchronize, the maximum over the ranks; pairs/s over the cells of ALL ranks."""
    import torch

    from similari_amd import sharding
    from similari_amd.engine import Engine

    keys = scene_ids_of_rank(total_scenes, world, rank)
    cfg, scenes, desc = workload(wname, seed=1234, scene_ids=keys)
    cfg.device = local_rank
    cfg.flags = 0
    eng = Engine(cfg)
    keep = stage(eng, cfg, scenes, keys)
    visual = cfg.visual_kind != abi.SA_VIS_NONE
    eng.batch_run()
    eng.batch_sync()
    got = [eng.batch_fetch(s, len(sc["det_boxes"])) for s, sc in enumerate(scenes)]
    cdev = "cpu" if dist.get_backend() == "gloo" else "cuda"
    tot = torch.tensor([float(sum(len(s["det_boxes"]) * len(s["track_boxes"]) for s in scenes)), float(sum(len(s["det_boxes"]) for s in scenes))], dtype=torch.float64, device=cdev)
    rows_all = [torch.zeros_like(tot) for _ in range(world)]
    dist.all_gather(rows_all, tot)
    total_cells = float(sum(float(t[0]) for t in rows_all))
    rows_per_rank = [int(t[1]) for t in rows_all]
    out = {"workload": desc, "scenes_total": total_scenes, "ranks": world, "backend": dist.get_backend(), "pairs_per_step": total_cells}
    for name, ondev in (([("device_features", True)] if visual else []) + [("host_pinned" if visual else "host_boxes", False)]):
        rg = sharding.ResultGather(max(rows_per_rank) + 16)

        def region(fn):
            barrier()
            t0 = time.perf_counter()
            fn()
            rg.drain()
            barrier()
            return max_over_ranks(time.perf_counter() - t0)

        res, ids = h2d_pipelined(eng, cfg, scenes, iters, feats_on_device=ondev, keys=keys, on_results=rg.push, region=region)
        rg.drain()
        same_local = all(np.array_equal(a, g[0]) for a, g in zip(ids, got))
        mine = np.concatenate([g[0] for g in got]) if got else np.zeros(0, np.uint64)
        everyone = [None] * world if rank == 0 else None
        dist.gather_object(mine, everyone, dst=0)      # (verification only, outside the timed region)
        ok = torch.tensor([1.0 if same_local else 0.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        entry = {"pairs_per_s": total_cells * iters / (1e-3 * res["ms_per_step"] * iters), "ms_per_step": res["ms_per_step"], "steps": iters,
                 "h2d_bytes_per_step_per_rank": res["h2d_bytes_per_step"], "gathered_bytes_per_step": 9 * sum(rows_per_rank),
                 "every_rank_matches_its_resident_run": bool(float(ok.item()) == 1.0)}
        if rank == 0:
            last = rg.last(rows_per_rank)
            entry["root_received_every_ranks_answers"] = bool(all(np.array_equal(last[r][0], everyone[r]) for r in range(world)))
        out[name] = entry
    eng.close()
    return out


def timed_rounds(run_k, barrier, min_total_s=0.5, min_rounds=5, max_rounds=400, max_over_ranks=None, after=None):
    """EXACTLY K steps per timed region, bracketed by barrier + synchronize on both sides; the region is repeated until at least
    min_total_s has been measured (a 20-step region of a 25 us step is 0.5 ms of signal: one region is a noisy sample) and the
    MEDIAN region is reported.  Under several ranks a region's time is the MAX over ranks (max_over_ranks: an all-reduce OUTSIDE the
    timed region) — the contract's definition, and what makes every rank take the same "one more region?" decision: with each rank
    summing its own clock, two ranks a microsecond apart at the threshold would leave the loop in different rounds and the next
    collective would never match."""
    times = []
    total = 0.0
    while len(times) < min_rounds or (total < min_total_s and len(times) < max_rounds):
        barrier()
        t0 = time.perf_counter()
        run_k()
        barrier()
        dt = time.perf_counter() - t0
        if after is not None:
            after()
        if max_over_ranks is not None:
            dt = max_over_ranks(dt)
        times.append(dt)
        total += dt
    return float(np.median(times)), times


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the timed CPU baselines (the oracle still checks the timed run's answer)")
    ap.add_argument("--no-oracle", action="store_true", help="skip match_vs_oracle as well")
    ap.add_argument("--profile-iters", type=int, default=50)
    ap.add_argument("--flags", type=int, default=-1, help="engine flags for the timed pass (default 0: eager launches, heterogeneous first phase where it applies; 8 hipGraph replay; 32 contraction as a kernel of its own)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe-inclusive pass (value_h2d)")
    ap.add_argument("--cluster", type=int, default=0, help="single process: also time the workload's scenes through sa_cluster over this many shards (one engine per device; --cluster-devices to place several shards on one GPU)")
    ap.add_argument("--cluster-devices", default="", help="comma-separated HIP ordinals for --cluster (default 0..n-1)")
    ap.add_argument("--gemm-plan", type=int, default=-1, help="pin the contraction's tile plan (sa_config.gemm_plan; tuning / A-B measurements)")
    ap.add_argument("--ingest", default="both", choices=["local", "scatter", "both"],
                    help="--gpus N > 1 on a scene set: `local` = every rank ingests the detections of ITS scenes, only ids / votes are gathered (this is `value`); "
                         "`scatter` = rank 0 packs and scatters the whole request set (value_scatter); default: both")
    ap.add_argument("--scenes", type=int, default=0, help="scene-set workloads (c2b, c2bk3, c3): scenes of the set — in all, split scene_id %% N under --gpus N (default 64 there, 8 at one GPU)")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SA_BENCH_FORCE_DIST=1: join a process group even at world size 1 (the RCCL branch — communicator, scatter / gather of the
    # dispatch pass — exercised on a one-GPU box: tests/test_gpu_cluster.py)
    if world > 1 or os.environ.get("SA_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if os.environ.get("SA_BENCH_ONE_DEVICE"):
            # rehearsal of the multi-rank control flow on a one-GPU box: every rank drives device 0, collectives over gloo
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        # one rank per GPU, each on its own share of the host's cores (pinned staging rows and pool threads on the rank's socket,
        # no two ranks' workers on one core): similari_amd.sharding.cpu_share
        from similari_amd import sharding as _sh

        rank_cpus = _sh.bind_rank_to_cpu_share(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    else:
        dist = None
        rank_cpus = []
        torch.cuda.set_device(local_rank)
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    from similari_amd.engine import Engine

    # --gpus N > 1 on a scene-set workload (and on the default one, whose batched form is c2b): a FIXED set of scenes, split scene_id % N
    wname = "c2b" if (args.workload == "c2" and world > 1) else args.workload
    fixed_set = world > 1 and wname in SCENE_SETS
    total_scenes = (args.scenes or 64) if fixed_set else (args.scenes or SCENE_SETS.get(wname, 0))
    if fixed_set:
        scene_keys = scene_ids_of_rank(total_scenes, world, rank)
        cfg, scenes, desc = workload(wname, seed=1234, scene_ids=scene_keys)
    else:
        replica_ids = list(range(args.scenes)) if (args.scenes and wname in SCENE_SETS) else None   # (None: the workload's default set)
        cfg, scenes, desc = workload(wname, seed=1234 + rank, scene_ids=replica_ids)
        scene_keys = list(range(len(scenes))) if scenes else []
    facade = None
    if cfg is None:  # c3m: tracks with Kalman states built by the product itself
        facade, eng, cells, acc0 = maha_engine(local_rank, 1234 + rank)
    else:
        cfg.device = local_rank
        cfg.flags = DEFAULT_FLAGS if args.flags < 0 else args.flags
        if args.gemm_plan >= 0:
            cfg.gemm_plan = args.gemm_plan + 1
        eng = Engine(cfg)
        keep = stage(eng, cfg, scenes, scene_keys)
        cells = sum(len(s["det_boxes"]) * len(s["track_boxes"]) for s in scenes)

    def barrier():
        # the contract's bracket: a barrier + torch.cuda.synchronize().  One rank: the synchronize alone (it drains every stream of the
        # device, the engine's included) — each further runtime call on the idle device is 2.7 us INSIDE the region, 0.13 us per step of a
        # 20-step region (scripts/region_fixed_cost.py: batch_sync + two synchronizes 20.8 us per step at K = 20, one synchronize 20.3)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def run_k():
        for _ in range(args.steps):
            eng.batch_run()

    def after_region():   # (outside the clock: the engine's own view of the drained queue — error codes, replaced buffers)
        eng.batch_sync()

    for _ in range(args.warmup):
        eng.batch_run()
    eng.batch_sync()
    dt, regions = timed_rounds(run_k, barrier, max_over_ranks=max_over_ranks, after=after_region)
    # the pure per-step time: a region carries a fixed cost (the first launch's latency, the final synchronisation: ~20 us, 5 % of a
    # 20-step region) — regions of 4 K steps give the slope; `value` stays the K-step region, the slope only rescales the
    # instrumented per-kernel durations below
    def run_4k():
        for _ in range(4 * args.steps):
            eng.batch_run()
    if dt < 0.05:
        dt4, _r4 = timed_rounds(run_4k, barrier, min_total_s=0.2, min_rounds=3, max_rounds=60, max_over_ranks=max_over_ranks, after=after_region)
        step_s = max(0.0, (dt4 - dt) / (3.0 * args.steps)) or dt / args.steps
    else:
        step_s = dt / args.steps  # a region of 50 ms and more: the fixed cost is below a tenth of a percent
    # per-kernel durations: hipEvents stamped with each dispatch's own begin / end on the engine's stream (hipExtLaunchKernelGGL; the
    # events release to the DEVICE, like the plain pipeline's dispatches), on the SAME engine and staged inputs, right behind the timed
    # regions — the GPU in the state the timed loop left it in (a second engine after seconds of CPU-side work read 1 us more per launch
    # of the fused first phase: clocks and caches of an idle device) — and after them, so that the timed regions stay free of instrumentation
    # The pass runs as GROUPS of a few launches each (the engine accumulates per kernel: a group's mean is the finest grain it hands out):
    # the line reports the MEDIAN of the group means with their 10th / 90th percentile — one slow dispatch (another tenant's kernel, a
    # clock step) moves a mean over 50 launches by percents, the median not at all.
    prof_groups = {}
    def profile_pass():
        eng.profile_enable(True)
        for _ in range(5):
            eng.batch_run()
        eng.batch_sync()
        total = {}
        per = 5
        for _g in range(max(1, args.profile_iters // per)):
            eng.profile_reset()
            for _ in range(per):
                eng.batch_run()
            eng.batch_sync()
            for k, (n, ms) in eng.profile_read().items():
                t = total.setdefault(k, [0, 0.0])
                t[0] += n; t[1] += ms
                if n:
                    prof_groups.setdefault(k, []).append(1e3 * ms / n)
        eng.profile_enable(False)
        return {k: (v[0], v[1]) for k, v in total.items()}
    prof = profile_pass()
    args.profile_iters = max(1, args.profile_iters // 5) * 5
    if dist is not None:
        cdev = "cpu" if dist.get_backend() == "gloo" else "cuda"
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ct = torch.tensor([float(cells)], dtype=torch.float64, device=cdev)
        dist.all_reduce(ct, op=dist.ReduceOp.SUM)
        total_cells = float(ct.item())
    else:
        total_cells = float(cells)
    # the timed work produced the right answer: against the synthetic truth, and (below, rank 0 at N = 1) against the oracle
    if facade is None:
        got = [eng.batch_fetch(s, len(sc["det_boxes"])) for s, sc in enumerate(scenes)]
        acc = float(np.mean(np.concatenate([g[0] == sc["truth"] for g, sc in zip(got, scenes)])))
    else:
        got = None
        ids, votes = eng.batch_fetch(0, 500)
        acc = float((ids != 0).mean())  # every detection of the staged frame continues a track

    # PCIe-inclusive: the same frame from host buffers every step, pipelined (detections H2D and assignments D2H included)
    h2d, h2d_ids = None, None
    if not args.no_h2d and facade is None:
        h2d, h2d_ids = h2d_pipelined(eng, cfg, scenes, max(args.steps, 50), keys=scene_keys)
        if dist is not None:
            th = torch.tensor([h2d["ms_per_step"]], dtype=torch.float64, device=cdev)
            dist.all_reduce(th, op=dist.ReduceOp.MAX)
            h2d["ms_per_step"] = float(th.item())
            h2d["pairs_per_s"] = total_cells / (1e-3 * h2d["ms_per_step"])
        h2d["matches_resident_run"] = bool(all(np.array_equal(a, g[0]) for a, g in zip(h2d_ids, got)))
    # the same with the feature rows already in HBM (the detector's ReID head ran on this GPU): only boxes + qualities cross PCIe
    devf = None
    if not args.no_h2d and facade is None and cfg.visual_kind != abi.SA_VIS_NONE and dist is None:
        devf, devf_ids = h2d_pipelined(eng, cfg, scenes, max(args.steps, 50), feats_on_device=True, keys=scene_keys)
        devf["matches_resident_run"] = bool(all(np.array_equal(a, g[0]) for a, g in zip(devf_ids, got)))

    # (the per-kernel durations were taken right after the timed regions, on the same engine: see profile_pass above)
    if facade is not None:
        eng.close()
        facade.close()
    else:
        eng.close()

    # N > 1: the request set of ALL ranks from ONE ingest point (rank 0) through the scene scatter / result gather north_star names
    # (similari_amd.sharding.ShardedAssociator: one scatter of packed shares, one sa_associate_batch per rank, one gather), inside
    # the timed region: W untimed request sets, then EXACTLY K of them between two barriers on rank 0's clock (every step ends with
    # the gather, so rank 0's clock is the maximum over the ranks).  On a FIXED scene set (scene-set workloads: the 64 scenes split
    # scene_id % N) this is `value` — total cells / wall time of scatter + per-rank batch + gather, strong scaling — and the per-rank
    # resident replay above becomes value_resident; on the other workloads (every rank replays its own frame: weak scaling) it stays
    # the side object `dispatch`.
    local = None
    local_c3 = None
    if dist is not None and facade is None and fixed_set and args.ingest in ("local", "both"):
        # (EXACTLY K steps between the barriers, as for every timed region of this file)
        local = local_ingest(wname, total_scenes, world, rank, local_rank, dist, barrier, max_over_ranks, args.steps)
        if wname == "c2b":   # BASELINE's multi-GPU configuration beside the headline one: 64 scenes x 500 x 500 BatchSORT (KB-scale ingest)
            local_c3 = local_ingest("c3", total_scenes, world, rank, local_rank, dist, barrier, max_over_ranks, args.steps)
    dispatch = None
    if dist is not None and facade is None and (args.ingest in ("scatter", "both") or not fixed_set):
        from similari_amd import sharding
        from similari_amd.engine import Engine as _E

        visual = cfg.visual_kind != abi.SA_VIS_NONE
        cfg.flags = 0
        deng = _E(cfg)
        if fixed_set:
            gids = scene_keys                                            # this rank's scenes of the fixed set (owner = id % world)
            everyone = [(sid, sc) for sid, sc in zip(range(total_scenes), workload(wname, seed=1234, scene_ids=list(range(total_scenes)))[1])] if rank == 0 else None
        else:
            gids = [rank + world * s_ for s_ in range(len(scenes))]      # replicas: global scene id = rank + world * local index -> owner = rank
            everyone = [(r + world * s_, sc) for r in range(world) for s_, sc in enumerate(workload(wname, seed=1234 + r, scene_ids=replica_ids)[1])] if rank == 0 else None
        # capacities: the largest share any rank receives (the set is split evenly up to one scene)
        per_scene_rows = max(len(sc["det_boxes"]) for sc in scenes) if scenes else 1
        n_mine = (total_scenes + world - 1) // world if fixed_set else len(scenes)
        rows = n_mine * per_scene_rows
        bytes_ = rows * (4 * cfg.feature_len if visual else 0) + 4096
        sh = sharding.ShardedAssociator(deng, capacity_bytes=bytes_, capacity_rows=rows + 16, max_scenes=max(64, n_mine))
        for gid, sc in zip(gids, scenes):  # every rank seeds its own scenes' tables
            kw = dict(feats=sc["track_feats"], feat_present=sc["track_present"]) if visual else {}
            deng.upsert(gid, abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"], **kw))
        barrier()
        if rank == 0:
            items = [(gid, 1, sc["det_boxes"], sc["det_feats"] if visual else None, sc["det_quality"] if visual else None) for gid, sc in everyone]
            for _ in range(max(1, min(args.warmup, 5))):
                res = sh.associate(items)
            iters = args.steps if fixed_set else min(args.steps, 30)
            t0 = time.perf_counter()
            for _ in range(iters):
                res = sh.associate(items)
            dtd = (time.perf_counter() - t0) / iters
            mine_at = [i for i, (gid, _) in enumerate(everyone) if gid % world == 0]  # rank 0's own scenes, in its staging order
            ok = all(np.array_equal(res[i][0], g[0]) for i, g in zip(mine_at, got))
            truth_ok = float(np.mean(np.concatenate([res[i][0] == sc["truth"] for i, (_, sc) in enumerate(everyone)])))
            sh.shutdown()
            dispatch = {"pairs_per_s": total_cells / dtd, "ms_per_batch": 1e3 * dtd, "steps": iters, "scenes": len(items), "ranks": world,
                        "backend": dist.get_backend(), "rank0_local_ms": sh.last_local_ms, "rank0_answers_match_resident_run": bool(ok),
                        "match_accuracy_all_scenes": truth_ok,
                        "note": "rank 0 packs every rank's share (numpy concatenation), ONE scatter, every rank runs one sa_associate_batch on its GPU "
                                "(H2D + kernels + results), ONE gather; host buffers in, host buffers out on rank 0"}
        else:
            sh.serve_forever()
        deng.close()
        barrier()

    # the scenes of this workload through the in-process dispatcher (sa_cluster): one ingest point, scatter by scene_id % shards,
    # every shard's share on its own engine concurrently, gather — host buffers in, host buffers out
    cluster = None
    if args.cluster and rank == 0 and facade is None:
        from similari_amd.engine import Cluster

        devs = [int(x) for x in args.cluster_devices.split(",")] if args.cluster_devices else list(range(args.cluster))
        visual = cfg.visual_kind != abi.SA_VIS_NONE
        cfg.flags = 0
        cl = Cluster(cfg, devices=devs)
        many = scenes * max(1, (len(devs) * max(1, 8 // max(1, len(scenes)))))  # every shard gets as many scenes as one GPU had
        items = []
        for s, sc in enumerate(many):
            kw = dict(feats=sc["track_feats"], feat_present=sc["track_present"]) if visual else {}
            cl.upsert(s, abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"], **kw))
            kw = dict(feats=sc["det_feats"], feat_quality=sc["det_quality"]) if visual else {}
            items.append((s, 1, abi.make_detections(sc["det_boxes"], **kw)))
        req, res, outs = Engine.make_requests(items)
        for _ in range(5):
            cl.associate_batch(req, res)
        t0 = time.perf_counter()
        shard_ms = np.zeros(len(devs))
        for _ in range(50):
            cl.associate_batch(req, res)
            shard_ms += np.array(cl.last_ms())
        dtc = (time.perf_counter() - t0) / 50
        ccells = sum(len(sc["det_boxes"]) * len(sc["track_boxes"]) for sc in many)
        cluster = {"shards": len(devs), "devices": devs, "scenes": len(many), "pairs_per_s": ccells / dtc, "ms_per_batch": 1e3 * dtc,
                   "shard_ms_per_batch": [float(x) / 50 for x in shard_ms],
                   "note": "sa_cluster_associate_batch from host buffers (pageable): split by scene_id % shards, one sa_associate_batch per shard on its "
                           "worker thread, all shards concurrently, results in the caller's arrays; shard_ms = wall time inside each worker"}
        cl.close()

    if rank == 0:
        # Per-kernel durations come from an INSTRUMENTED pass: every launch carries two events stamped with the dispatch's own begin /
        # end (hipExtLaunchKernelGGL) — the clock rocprofv3's kernel trace reads.  avg_us = those durations AS MEASURED (a dispatch that
        # signals completion also ends with a system-scope release the same kernel inside the plain pipeline does not pay: ~1.5 us on
        # the fused first phase, so one step's launches may add up to slightly MORE than ms_per_step); avg_us_rocprof = the average
        # rocprofv3 --kernel-trace --stats reported for the same command, from the summary committed under profiles/ (None when there is none).
        # avg_us = the MEDIAN over the groups of the instrumented pass (mean_us: the plain mean over every launch, p10_us / p90_us: the groups' spread)
        raw = {k: (float(np.median(prof_groups[k])) if prof_groups.get(k) else 1e3 * ms / max(n, 1)) for k, (n, ms) in prof.items()}
        rp = rocprof_stats(args.workload)
        kern = {k: {"launches": int(prof[k][0]), "avg_us": raw[k], "mean_us": 1e3 * prof[k][1] / max(prof[k][0], 1),
                    "p10_us": float(np.percentile(prof_groups[k], 10)) if prof_groups.get(k) else None,
                    "p90_us": float(np.percentile(prof_groups[k], 90)) if prof_groups.get(k) else None,
                    "avg_us_rocprof": rp.get(k)} for k in raw}
        gpu_kernels = {k: v for k, v in kern.items() if k != "d2h_results"}
        visual = facade is None and cfg.visual_kind != abi.SA_VIS_NONE
        per_step = lambda k: prof[k][0] / float(args.profile_iters)  # launches of kernel k per step
        # ---- algorithmic work per launch, stated in DESIGN.md section 4 ----
        models = {}
        if visual:
            euclid = cfg.visual_kind == abi.SA_VIS_EUCLIDEAN
            # euclidean engines run on the matrix cores too (expansion + flagged direct recompute) unless told otherwise or the
            # feature length rules it out (rho = 5e-3 sqrt(Dp) >= 1/3): then sub, mul, add per element on the vector pipe
            eu_valu = euclid and (bool(cfg.flags & abi.SA_FLAG_EUCLID_VALU) or 5e-3 * ((cfg.feature_len + 31) // 32 * 32) ** 0.5 >= 1.0 / 3.0)
            K = cfg.max_observations
            flops = sum((3.0 if eu_valu else 2.0) * len(s["det_boxes"]) * len(s["track_boxes"]) * K * cfg.feature_len for s in scenes)
            models["k_visual_cost"] = ("valu" if eu_valu else "mfma", flops)
            models["k_frame_visual"] = ("mfma", flops)
            models["k_visual_raw"] = ("mfma", flops)   # the contraction on the raw rows, running beside k_frame
            models["k_bestfit_tile"] = ("hbm", 4.0 * K * cells + 12.0 * (cells / 64.0) * 2.0)
        pairs_near, flop_pair = (0, 0.0)
        if facade is None:
            pairs_near, flop_pair = f64_clip_model(scenes)
            edges = sum(int((g[0] != 0).sum()) for g in got) if got else 0
            models["k_frame"] = ("hbm", kernel_bytes_model(cfg, scenes, edges) + (sum(len(s["det_boxes"]) for s in scenes) * 4.0 * cfg.feature_len * 2 if visual else 0.0))
        else:
            models["k_frame"] = ("hbm", (48.0 + 84.0 + 80.0 + 16.0) * 8 * 500)
        dom = max(gpu_kernels, key=lambda k: gpu_kernels[k]["avg_us"] * gpu_kernels[k]["launches"])
        roof = None
        if dom in models:
            bound, amount = models[dom]
            per_launch = amount / per_step(dom)
            dur_s = kern[dom]["avg_us"] * 1e-6
            if bound == "valu":
                a = per_launch / dur_s / 1e12
                roof = {"kernel": dom, "bound": "valu", "achieved": a, "peak": VALU_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": a / VALU_F32_PEAK_TFLOPS, "traffic": None,
                        "peak_note": "f32 vector peak (the same 157.3 TFLOP/s as the f32 matrix cores); sub + mul + add per element, 2 of 3 fuse"}
            elif bound == "mfma":
                a = per_launch / dur_s / 1e12
                roof = {"kernel": dom, "bound": "mfma", "achieved": a, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": a / MFMA_F32_PEAK_TFLOPS, "traffic": None}
            else:
                a = per_launch / dur_s / 1e9
                roof = {"kernel": dom, "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": a / HBM_PEAK_GBS, "traffic": None}
                if dom == "k_frame":
                    roof["note"] = ("the positional launch is an elementwise map, HBM-bound by SURVEY 8(d)'s definition, but it moves kilobytes per microsecond: "
                                    "what bounds it is the f64 vector work of the clip (valu_f64 below) and the latency chain of a tile")
        else:
            roof = {"kernel": dom, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                    "traffic": None, "note": "latency-bound helper kernel; no algorithmic-byte model"}
        if roof is not None and dom in models and kern[dom].get("avg_us_rocprof"):
            # the same figure from the committed rocprofv3 average of this kernel (no per-dispatch completion signal in that run:
            # the instrumented duration above carries ~1.5 us of it on the fused first phase)
            per_launch = models[dom][1] / per_step(dom)
            rate = per_launch / (kern[dom]["avg_us_rocprof"] * 1e-6)
            roof["achieved_rocprof"] = rate / (1e9 if roof["unit"] == "GB/s" else 1e12)
            roof["frac_rocprof"] = roof["achieved_rocprof"] / roof["peak"]
            roof["rocprof_source"] = "profiles/r*_%s_kernel_stats.csv (newest round)" % args.workload
        if roof is not None:
            tr = pmc_traffic(args.workload, dom)
            if tr:
                roof["traffic"], roof["traffic_source"] = tr[0], f"profiles/{tr[1]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 2*FETCH + WRITE bytes per launch)"
                if tr[2]:
                    roof["traffic_stale"] = "measured in an earlier round than the newest committed PMC pass: not this tree's launch"
            if dom in models:
                roof["algorithmic"] = models[dom][1] / per_step(dom)
                roof["algorithmic_unit"] = "flop per launch" if models[dom][0] in ("mfma", "valu") else "bytes per launch"
        # secondary roofline lines for every modelled kernel
        for k, (bound, amount) in models.items():
            if k in kern and kern[k]["avg_us"] > 0:
                per_launch = amount / per_step(k)
                rate = per_launch / (kern[k]["avg_us"] * 1e-6)
                mp = MFMA_F32_PEAK_TFLOPS
                kern[k]["roofline_frac"] = rate / 1e12 / mp if bound == "mfma" else rate / 1e12 / VALU_F32_PEAK_TFLOPS if bound == "valu" else rate / 1e9 / HBM_PEAK_GBS
        # the positional tiles' f64 vector work (they run inside k_frame, or inside k_frame_visual beside the contraction)
        pk = "k_frame" if "k_frame" in kern else ("k_frame_visual" if "k_frame_visual" in kern else None)
        valu_f64 = None
        if pk and pairs_near:
            fl = pairs_near * flop_pair / per_step(pk)
            valu_f64 = {"kernel": pk, "surviving_pairs_per_launch": pairs_near / per_step(pk), "f64_flop_per_pair": flop_pair, "f64_flop_per_launch": fl,
                        "achieved": fl / (kern[pk]["avg_us"] * 1e-6) / 1e12, "peak": VALU_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": fl / (kern[pk]["avg_us"] * 1e-6) / 1e12 / VALU_F64_PEAK_TFLOPS,
                        "note": "pairs that pass too_far() x the f64 operations Sutherland-Hodgman + shoelace spend on them (counted on a sample by a host restatement)"}
        lead = None   # N > 1 on a fixed scene set: the ranks ingest their own scenes (features resident in HBM where the workload has any)
        if local is not None:
            lead = local.get("device_features") or local.get("host_boxes") or local.get("host_pinned")
        out = {
            "metric": "assoc-pairs/sec (NxM cost+assign) VisualSORT 512-d",
            "value": (lead["pairs_per_s"] if lead else dispatch["pairs_per_s"] if (fixed_set and dispatch) else total_cells * args.steps / dt),
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": (lead["ms_per_step"] if lead else dispatch["ms_per_batch"] if (fixed_set and dispatch) else 1e3 * dt / args.steps),
            "higher_is_better": True,
            "scaling": "strong" if fixed_set else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": (("synthetic (seeded, SURVEY §8d); a FIXED set of %d scenes split scene_id %% %d; every rank ingests the detections of ITS scenes each step (boxes from "
                      "host buffers, feature rows resident in HBM), runs them, ids / votes gathered on rank 0 (RCCL) — value = total cells / that wall time (maximum over "
                      "ranks); value_h2d = the same with the feature rows in pinned host blocks; value_scatter = rank 0 packs and scatters the whole request set; "
                      "value_resident = the per-rank replay with inputs resident in HBM" if lead else
                      "synthetic (seeded, SURVEY §8d); a FIXED set of %d scenes split scene_id %% %d: every step rank 0 scatters the request set (RCCL), every rank "
                      "runs its share (H2D + kernels), one gather — value = total cells / that wall time; value_resident = the per-rank replay with inputs resident in HBM")
                     % (total_scenes, world)) if fixed_set else "synthetic (seeded, SURVEY §8d), inputs resident in HBM, same frame replayed each step",
            "config": {"workload": desc, "scenes_per_gpu": len(scenes) if scenes else 8, "scenes_total": (total_scenes if fixed_set else world * (len(scenes) if scenes else 8)),
                       "pairs_per_step_per_gpu": cells,
                       "parallelism": ((f"fixed scene set sharded scene_id % {world}, every rank ingests its own scenes: one KB-scale gather of ids / votes per step (RCCL), "
                                        "no collective inside a rank's share" if lead else
                                        f"fixed scene set sharded scene_id % {world}: one scatter + one gather per step (RCCL), no collective inside a rank's share")
                                       if fixed_set else f"scene-sharded x{world}, no data-path collective")},
            "timed_regions": {"count": len(regions), "steps_each": args.steps, "reported": "median", "min_ms_per_step": 1e3 * min(regions) / args.steps,
                              "max_ms_per_step": 1e3 * max(regions) / args.steps, "ms_per_step_slope": 1e3 * step_s},
            "value_resident": total_cells * args.steps / dt,
            "ms_per_step_resident": 1e3 * dt / args.steps,
            "value_h2d": (local["host_pinned"]["pairs_per_s"] if (local and "host_pinned" in local) else h2d["pairs_per_s"] if h2d else None),
            "value_scatter": dispatch["pairs_per_s"] if (fixed_set and dispatch) else None,
            "match_accuracy": acc,
            "roofline": roof,
            "kernels": kern,
        }
        if valu_f64:
            out["valu_f64"] = valu_f64
        # Frames whose launches are chains of latencies, not streams (positional-only workloads: the dominant kernel has no matrix-core
        # model and moves kilobytes): an HBM fraction of 0.005 says nothing — what such a frame costs is read against the floor of its
        # dependent launches (an empty kernel in a chain: ~3 us each on this stack)
        lf = launch_floor_us()
        if lf is not None and roof is not None and roof.get("bound") == "hbm":
            n_launch = sum(per_step(k) for k in gpu_kernels)
            out["fixed_cost"] = {"frame_us": 1e6 * dt / args.steps, "launches_per_frame": n_launch, "launch_floor_us": lf[0],
                                 "floor_of_the_frame_us": n_launch * lf[0], "above_the_floor_us": 1e6 * dt / args.steps - n_launch * lf[0],
                                 "source": f"profiles/{lf[1]} (scripts/micro/launch_floor.hip: an empty 1000-block kernel, average of a chain of 200 dependent launches)",
                                 "note": "positional launches and assignment tails are chains of dependent round trips (in-kernel timelines: profiles/r04_z_pos_trace.txt); "
                                         "read the frame against its launch floor, not against HBM bandwidth"}
        if h2d is not None:
            out["h2d_inclusive"] = h2d
        if devf is not None:
            out["device_features_inclusive"] = devf
        if cluster is not None:
            out["cluster"] = cluster
        if dispatch is not None:
            out["dispatch"] = dispatch
        if world > 1:
            out["cpus_of_rank0"] = len(rank_cpus)   # (0: the process was left where the launcher put it)
        if local is not None:
            out["ingest_local"] = local
        if local_c3 is not None:
            out["c3_batchsort"] = local_c3
        if not args.no_oracle and world == 1 and facade is None:  # the timed run's answer against the oracle's (ids AND vote types)
            ans = oracle_answers(cfg, scenes)
            same = np.concatenate([(g[0] == a[0]) & (g[1] == a[1]) for g, a in zip(got, ans)])
            out["match_vs_oracle"] = float(same.mean())
        if world == 1 and facade is None and args.workload in ("giant", "bigpile", "bigcrowd", "sd", "sdt"):
            try:
                out["cpu_solve_only"] = cpu_solve_only(cfg, scenes)
            except Exception as ex:
                out["cpu_solve_only"] = {"error": repr(ex)}
        if not args.no_cpu_baseline and world == 1 and facade is None:  # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(cfg, scenes)
            try:
                out["cpu_baseline_threads"] = cpu_baseline_threads(cfg, scenes)
            except Exception as ex:  # the threaded variant is an extra: never lose the bench line over it
                out["cpu_baseline_threads"] = {"error": repr(ex)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
