#!/usr/bin/env python
"""bench.py — assoc-pairs/sec of the association hot path on MI355X (BASELINE.json metric).

A step = one pass of the hot path (pair pre-filter + cost matrices + quantise + assignment) over one scene-frame
whose inputs are already resident in HBM.  At N GPUs every rank owns its own scene (scenes are independent:
compatible() is false across scene ids, sort.rs:251), so there is no data-path collective and scaling is weak.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5]
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
MFMA_F16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_f16 dense peak (MI355X_MICROARCH.md); the f16-split option issues 3 products per flop


def workload(name: str, seed: int):
    """Returns (config, scene dict, description).  C2 is the configuration the metric is quoted on."""
    rng = np.random.default_rng(seed)
    if name == "c2":
        n = t = 1000
        d, k = 512, 1
        sc = synth.visual_scene(rng, t, n, d, k)
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual="cosine", visual_threshold=0.2, feature_len=d,
                              max_observations=k, visual_min_votes=1, visual_minimal_track_length=1,
                              positional_min_confidence=0.1, max_idle_epochs=5)
        return cfg, [sc], "VisualSORT 1000 tracks x 1000 dets, 512-d cosine + IoU(0.3), K=1 (BASELINE C2)"
    if name == "c2k3":
        n = t = 1000
        d, k = 512, 3
        sc = synth.visual_scene(rng, t, n, d, k)
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual="cosine", visual_threshold=0.2, feature_len=d,
                              max_observations=k, visual_min_votes=1, visual_minimal_track_length=1,
                              positional_min_confidence=0.1, max_idle_epochs=5)
        return cfg, [sc], "VisualSORT 1000 tracks x 1000 dets, 512-d cosine + IoU(0.3), K=3 observations per track (BASELINE C2, deeper bank)"
    if name == "c2e":
        n = t = 1000
        d, k = 512, 1
        sc = synth.visual_scene(rng, t, n, d, k)
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, visual="euclidean", visual_threshold=0.5, feature_len=d,
                              max_observations=k, visual_min_votes=1, visual_minimal_track_length=1,
                              positional_min_confidence=0.1, max_idle_epochs=5)
        return cfg, [sc], "VisualSORT 1000 tracks x 1000 dets, 512-d EUCLIDEAN + IoU(0.3), K=1 (the C2 frame with the other visual metric)"
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
        scs = [synth.sort_scene(rng, 500, 500, canvas=(4096.0, 4096.0)) for _ in range(8)]
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
        return cfg, scs, "BatchSORT IoU, 8 scenes x 500 x 500 per GPU (BASELINE C3: 64 scenes over 8 GPUs)"
    if name == "c1":
        sc = synth.sort_scene(rng, 100, 100, canvas=(1920.0, 1080.0))
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
        return cfg, [sc], "SORT IoU 100 x 100, dense variant (BASELINE C1)"
    if name == "sd":
        # not a BASELINE config: a crowd for plain SORT (the C2 frame without features) — the positional vote alone has to untangle
        # the overlaps, so its graph has large connected components
        sc = synth.sort_scene(rng, 1000, 1000, canvas=(1920.0, 1080.0))
        cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
        return cfg, [sc], "SORT IoU 1000 x 1000 on the C2 canvas (crowd: large components in the positional vote)"
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


def stage(eng, cfg, scenes):
    visual = cfg.visual_kind != abi.SA_VIS_NONE
    keep = []
    for s, sc in enumerate(scenes):
        kw = dict(feats=sc["track_feats"], feat_present=sc["track_present"]) if visual else {}
        tr = abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"], **kw)
        eng.upsert(s, tr)
        keep.append(tr)
    eng.batch_begin()
    dets = []
    for s, sc in enumerate(scenes):
        kw = dict(feats=sc["det_feats"], feat_quality=sc["det_quality"]) if visual else {}
        d = abi.make_detections(sc["det_boxes"], **kw)
        eng.batch_add(s, 1, d)
        dets.append(d)
    return keep, dets


# Algorithmic work per launch of each kernel, as (bound, amount, unit-per-second divisor).
def kernel_models(cfg, scenes):
    visual = cfg.visual_kind != abi.SA_VIS_NONE
    K = cfg.max_observations if visual else 1
    D8 = (cfg.feature_len + 31) // 32 * 32
    cells = sum(len(s["det_boxes"]) * len(s["track_boxes"]) for s in scenes)
    nt = sum(len(s["det_boxes"]) + len(s["track_boxes"]) for s in scenes)
    n_ = sum(len(s["det_boxes"]) for s in scenes)
    t_ = sum(len(s["track_boxes"]) for s in scenes)
    # k_frame = positional tiles + frame-preparation blocks in one launch.  SURVEY §8d figure for the positional cells: f32 cost
    # out (4 B/cell) + 64 B vertices + 16 B geometry per box — effective bandwidth, the kernel no longer writes the dense matrix
    # (it emits the edges of the vote directly); plus what the preparation blocks really move (features in and out, 160 B/box).
    frame_bytes = 4.0 * cells + 80.0 * nt + 160.0 * n_
    m = {
        # one read of the visual weights (4 B per cell and bank slot) + the per-tile partials
        "k_bestfit_tile": ("hbm", 4.0 * K * cells + 12.0 * (cells / 64.0) * 2.0),
    }
    if visual:
        euclid = cfg.visual_kind == abi.SA_VIS_EUCLIDEAN
        # euclidean: sub, mul, add per element on the f32 vector pipe (no matrix-core form without catastrophic cancellation)
        flops = sum((3.0 if euclid else 2.0) * len(s["det_boxes"]) * len(s["track_boxes"]) * K * cfg.feature_len for s in scenes)
        m["k_visual_cost"] = ("valu" if euclid else "mfma", flops)
        # small frames: contraction tiles + positional tiles + preparation blocks in one heterogeneous launch — the matrix-core
        # work is what bounds it; the other two kinds fill the issue slots it leaves idle
        m["k_frame_visual"] = ("mfma", flops)
        frame_bytes += 4.0 * n_ * (cfg.feature_len + D8)
    m["k_frame"] = ("hbm", frame_bytes)
    return m, cells


def pmc_traffic(workload: str, kernel: str):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r01_*_pmc_traffic.json, written by
    scripts/pmc_traffic.sh on the GPU box: separate FETCH_SIZE / WRITE_SIZE passes, FETCH doubled as the gfx950 guide says).
    Counters cannot be read from inside this process; None when no pass exists for this workload."""
    best = None
    for f in sorted((ROOT / "profiles").glob("r*_pmc_traffic.json")):
        try:
            d = json.loads(f.read_text())["workloads"].get(workload, {}).get(kernel)
        except Exception:
            d = None
        if d:
            best = (d["hbm_bytes"], f.name)
    return best


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
        "sample": f"oracle or_associate (oracle/oracle.cpp, -O2 -ffp-contract=off), {n} of {n_all} detections x {T} tracks of scene 0, "
                  f"best of {len(times)} runs ({sum(times):.1f} s of CPU work, 1 thread; host has {os.cpu_count()} cores)",
    }


def cpu_baseline_threads(cfg, scenes, budget_s=6.0):
    """The same oracle on `shards` host threads, tracks partitioned track_id % shards the way the reference's TrackStore shards them
    (store.rs:490-493).  Every shard runs the whole per-frame path on its tracks (ctypes releases the GIL), votes included — the
    reference votes once, on one thread, after the shards have produced their distances — so this favours the CPU.  Reported
    beside cpu_baseline (BASELINE.md section 2 asks for both variants); it is not the judged baseline object."""
    import oracle_lib as O
    from concurrent.futures import ThreadPoolExecutor

    sc = scenes[0]
    visual = cfg.visual_kind != abi.SA_VIS_NONE
    T, N = len(sc["track_boxes"]), len(sc["det_boxes"])
    shards = int(max(1, min(os.cpu_count() or 1, 64, T)))
    parts = []
    for sh in range(shards):
        m = (sc["track_ids"] % np.uint64(shards)) == sh
        if not m.any():
            continue
        kw = dict(feats=sc["track_feats"][m], feat_present=sc["track_present"][m]) if visual else {}
        parts.append(abi.make_tracks(sc["track_ids"][m], sc["track_boxes"][m], sc["track_epochs"][m], **kw))
    kw = dict(feats=sc["det_feats"], feat_quality=sc["det_quality"]) if visual else {}
    det = abi.make_detections(sc["det_boxes"], **kw)

    def one(tr):
        O.associate(cfg, tr, 1, det, want_matrices=False)

    times = []
    with ThreadPoolExecutor(max_workers=len(parts)) as ex:
        t_end = time.perf_counter() + budget_s
        while len(times) < 2 or (time.perf_counter() < t_end and len(times) < 50):
            t0 = time.perf_counter()
            list(ex.map(one, parts))
            times.append(time.perf_counter() - t0)
    return {
        "value": N * T / min(times), "unit": "pairs/s", "cores": len(parts), "kind": "port",
        "sample": f"oracle or_associate on {len(parts)} threads, tracks of scene 0 partitioned id % {shards} (store.rs:490-493), all {N} detections x {T} "
                  f"tracks per frame, per-shard votes run in parallel (favours the CPU), best of {len(times)} frames; host has {os.cpu_count()} cores",
    }


def h2d_inclusive(eng, cfg, scenes, iters=30):
    """sa_associate from HOST buffers (stage + H2D + pipeline + results): the PCIe-inclusive rate.  Reported, never `value`."""
    visual = cfg.visual_kind != abi.SA_VIS_NONE
    dets = []
    for sc in scenes:
        kw = dict(feats=sc["det_feats"], feat_quality=sc["det_quality"]) if visual else {}
        dets.append(abi.make_detections(sc["det_boxes"], **kw))
    for _ in range(3):
        for s, d in enumerate(dets):
            eng.associate(s, 1, d)
    t0 = time.perf_counter()
    for _ in range(iters):
        for s, d in enumerate(dets):
            eng.associate(s, 1, d)
    dt = time.perf_counter() - t0
    cells = sum(len(s["det_boxes"]) * len(s["track_boxes"]) for s in scenes)
    out = {"pairs_per_s": cells * iters / dt, "ms_per_frame_set": 1e3 * dt / iters,
           "note": "sa_associate per scene from pageable host buffers, synchronous: staging copy + H2D + pipeline + result fetch"}
    if visual:
        # the same from pinned blocks (sa_host_alloc): the DMA reads the caller's features in place
        blocks, pdets = [], []
        for sc in scenes:
            b = eng.host_block(sc["det_feats"].shape)
            b[...] = sc["det_feats"]
            blocks.append(b)
            pdets.append(abi.make_detections(sc["det_boxes"], feats=b, feat_quality=sc["det_quality"]))
        for _ in range(3):
            for s, d in enumerate(pdets):
                eng.associate(s, 1, d)
        t0 = time.perf_counter()
        for _ in range(iters):
            for s, d in enumerate(pdets):
                eng.associate(s, 1, d)
        dt2 = time.perf_counter() - t0
        out["pinned"] = {"pairs_per_s": cells * iters / dt2, "ms_per_frame_set": 1e3 * dt2 / iters,
                         "note": "features in a block from sa_host_alloc: no staging copy"}
        pdets = None
        for b in blocks:
            eng.host_free(b)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-iters", type=int, default=50)
    ap.add_argument("--flags", type=int, default=-1, help="engine flags for the timed pass (default 0: eager launches, heterogeneous first phase where it applies; 8 hipGraph replay; 32 contraction as a kernel of its own; 64 f16-split operands on the f16 matrix cores)")
    ap.add_argument("--h2d", action="store_true", help="also report the PCIe-inclusive rate of sa_associate from host buffers")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("SA_BENCH_ONE_DEVICE"):
            # rehearsal of the multi-rank control flow on a one-GPU box: every rank drives device 0, collectives over gloo
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(local_rank)
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    from similari_amd.engine import Engine

    cfg, scenes, desc = workload(args.workload, seed=1234 + rank)
    facade = None
    if cfg is None:  # c3m: tracks with Kalman states built by the product itself
        facade, eng, cells, acc0 = maha_engine(local_rank, 1234 + rank)
        models = {"k_frame": ("hbm", 4.0 * cells + 80.0 * 2 * 8 * 500 + 160.0 * 8 * 500)}
    else:
        cfg.device = local_rank
        cfg.flags = DEFAULT_FLAGS if args.flags < 0 else args.flags
        eng = Engine(cfg)
        keep = stage(eng, cfg, scenes)
        models, cells = kernel_models(cfg, scenes)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.batch_run()
    eng.batch_sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.batch_run()
    eng.batch_sync()
    barrier()  # barrier + synchronize on both sides of the timed region: every rank's dt covers the slowest rank
    dt = time.perf_counter() - t0
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
    # sanity: the timed work produced the right answer
    if facade is None:
        ids, votes = eng.batch_fetch(0, len(scenes[0]["det_boxes"]))
        acc = float((ids == scenes[0]["truth"]).mean())
    else:
        ids, votes = eng.batch_fetch(0, 500)
        acc = float((ids != 0).mean())  # every detection of the staged frame continues a track

    h2d = h2d_inclusive(eng, cfg, scenes) if (args.h2d and rank == 0 and facade is None) else None
    # per-kernel durations: hipEvents stamped with each dispatch's own begin / end on the engine's stream (hipExtLaunchKernelGGL),
    # same staged inputs, separate pass so that the timed region above stays free of instrumentation
    eng.close()
    if facade is not None:
        facade.close()
        prof = {}
    else:
        cfg_p = cfg
        cfg_p.flags = abi.SA_FLAG_PROFILE | (cfg.flags & (abi.SA_FLAG_FUSED_FRAME | abi.SA_FLAG_SEPARATE_FRAME | abi.SA_FLAG_F16_SPLIT))  # same launches as the timed pass
        engp = Engine(cfg_p)
        keep2 = stage(engp, cfg_p, scenes)
        for _ in range(5):
            engp.batch_run()
        engp.batch_sync()
        engp.profile_reset()
        for _ in range(args.profile_iters):
            engp.batch_run()
        engp.batch_sync()
        prof = engp.profile_read()
        engp.close()

    if rank == 0:
        kern = {k: {"launches": int(n), "avg_us": 1e3 * ms / max(n, 1)} for k, (n, ms) in prof.items()}
        gpu_kernels = {k: v for k, v in kern.items() if k != "d2h_results"}
        if not gpu_kernels:  # c3m: the engine belongs to the facade, no instrumented second pass
            print(json.dumps({"metric": "assoc-pairs/sec (NxM cost+assign)", "value": total_cells * args.steps / dt, "unit": "pairs/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": desc, "scenes_per_gpu": 8, "pairs_per_step_per_gpu": cells}, "match_accuracy": acc,
                              "roofline": None}))
            if dist is not None:
                dist.barrier()
                dist.destroy_process_group()
            return
        dom = max(gpu_kernels, key=lambda k: gpu_kernels[k]["avg_us"] * gpu_kernels[k]["launches"])
        roof = None
        if dom in models:
            bound, amount = models[dom]
            per_launch = amount * (prof[dom][0] and args.profile_iters / prof[dom][0])
            dur_s = kern[dom]["avg_us"] * 1e-6
            if bound == "valu":
                a = per_launch / dur_s / 1e12
                roof = {"kernel": dom, "bound": "valu", "achieved": a, "peak": VALU_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": a / VALU_F32_PEAK_TFLOPS, "traffic": None,
                        "peak_note": "f32 vector peak (the same 157.3 TFLOP/s as the f32 matrix cores); sub + mul + add per element, 2 of 3 fuse"}
            elif bound == "mfma":
                a = per_launch / dur_s / 1e12
                mfma_peak = MFMA_F16_PEAK_TFLOPS / 3.0 if (cfg.flags & abi.SA_FLAG_F16_SPLIT) else MFMA_F32_PEAK_TFLOPS
                roof = {"kernel": dom, "bound": "mfma", "achieved": a, "peak": mfma_peak, "unit": "TFLOP/s",
                        "frac": a / mfma_peak, "traffic": None}
                if cfg.flags & abi.SA_FLAG_F16_SPLIT:
                    roof["peak_note"] = "f16 MFMA dense peak / 3: the f16-split contraction issues three f16 products per algorithmic product"
            else:
                a = per_launch / dur_s / 1e9
                roof = {"kernel": dom, "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": a / HBM_PEAK_GBS, "traffic": None}
        else:
            roof = {"kernel": dom, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                    "traffic": None, "note": "latency-bound helper kernel; no algorithmic-byte model"}
        if roof is not None:
            tr = pmc_traffic(args.workload, dom)
            if tr:
                roof["traffic"], roof["traffic_source"] = tr[0], f"profiles/{tr[1]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 2*FETCH + WRITE bytes per launch)"
            if dom in models:
                roof["algorithmic"] = models[dom][1] * args.profile_iters / prof[dom][0]
                roof["algorithmic_unit"] = "flop per launch" if models[dom][0] in ("mfma", "valu") else "bytes per launch"
        # secondary roofline lines for every modelled kernel
        for k, (bound, amount) in models.items():
            if k in kern and kern[k]["avg_us"] > 0:
                per_launch = amount * args.profile_iters / prof[k][0]
                rate = per_launch / (kern[k]["avg_us"] * 1e-6)
                mp = MFMA_F16_PEAK_TFLOPS / 3.0 if (cfg.flags & abi.SA_FLAG_F16_SPLIT) else MFMA_F32_PEAK_TFLOPS
                kern[k]["roofline_frac"] = rate / 1e12 / mp if bound == "mfma" else rate / 1e12 / VALU_F32_PEAK_TFLOPS if bound == "valu" else rate / 1e9 / HBM_PEAK_GBS
        out = {
            "metric": "assoc-pairs/sec (NxM cost+assign) VisualSORT 512-d",
            "value": total_cells * args.steps / dt,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16-split operands (22 bit), f32 accumulate — NOT f32 arithmetic" if (cfg.flags & abi.SA_FLAG_F16_SPLIT) else "f32",
            "data": "synthetic (seeded, SURVEY §8d), inputs resident in HBM, same frame replayed each step",
            "config": {"workload": desc, "scenes_per_gpu": len(scenes), "pairs_per_step_per_gpu": cells,
                       "parallelism": f"scene-sharded x{world}, no data-path collective"},
            "match_accuracy": acc,
            "roofline": roof,
            "kernels": kern,
        }
        if h2d is not None:
            out["h2d_inclusive"] = h2d
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
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
