"""The request-set entry points of the C ABI on the GPU: sa_associate_batch (BatchSort / BatchVisualSort::predict's seam,
sort/batch_api.rs:222-290), the pipelined sa_pipe_* (two request sets in flight: H2D of set n+1 beside the kernels of set n),
sa_batch_time, and hipGraph replay under a tracker's ever-changing epoch — every answer against the oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from similari_amd import abi, synth
from similari_amd.engine import Engine, EngineError

pytestmark = pytest.mark.gpu


def visual_cfg(d, k=1, **kw):
    return abi.make_config(positional="iou", positional_threshold=0.3, visual="cosine", visual_threshold=0.2, feature_len=d,
                           max_observations=k, visual_min_votes=1, visual_minimal_track_length=1, positional_min_confidence=0.1,
                           max_idle_epochs=5, **kw)


def upsert_scene(eng, scene, sc, visual):
    kw = dict(feats=sc["track_feats"], feat_present=sc["track_present"]) if visual else {}
    tr = abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"], **kw)
    eng.upsert(scene, tr)
    return tr


def test_associate_batch_matches_oracle_per_scene():
    """sa_associate_batch: ragged scenes (one empty) in one call; res[i] belongs to req[i]."""
    rng = np.random.default_rng(71)
    cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
    sizes = [(120, 100), (0, 40), (33, 65), (257, 200)]
    scs = [synth.sort_scene(rng, t, n, canvas=(1200.0, 900.0)) for n, t in sizes]
    eng = Engine(cfg)
    try:
        trs = [upsert_scene(eng, 10 + s, sc, False) for s, sc in enumerate(scs)]
        dets = [abi.make_detections(sc["det_boxes"]) for sc in scs]
        req, res, outs = Engine.make_requests([(10 + s, 1, d) for s, d in enumerate(dets)])
        eng.associate_batch(req, res)
        for s, sc in enumerate(scs):
            ref = O.associate(cfg, trs[s], 1, dets[s], want_matrices=False)
            np.testing.assert_array_equal(outs[s][0], ref["track_id"], err_msg=f"scene {s}")
            np.testing.assert_array_equal(outs[s][1], ref["voting_type"])
        # the same scene twice in one request set is a caller error, reported, not undefined behaviour
        req2, res2, _ = Engine.make_requests([(10, 1, dets[0]), (10, 1, dets[0])])
        with pytest.raises(EngineError) as ei:
            eng.associate_batch(req2, res2)
        assert ei.value.code == abi.SA_ERR_STATE
    finally:
        eng.close()


def test_batch_time_replays_the_staged_set():
    """sa_batch_time: hipEvent time of `iters` back-to-back pipelines over the staged set; results stay the oracle's."""
    rng = np.random.default_rng(72)
    d = 64
    cfg = visual_cfg(d)
    sc = synth.visual_scene(rng, 200, 180, d, 1, canvas=(1500.0, 900.0), new_fraction=0.1)
    eng = Engine(cfg)
    try:
        tr = upsert_scene(eng, 0, sc, True)
        det = abi.make_detections(sc["det_boxes"], feats=sc["det_feats"], feat_quality=sc["det_quality"])
        eng.batch_begin()
        slot = eng.batch_add(0, 1, det)
        ms1 = eng.batch_time(1)    # includes the one-time upload of the staged set
        ms20 = eng.batch_time(20)
        ms200 = eng.batch_time(200)
        assert ms1 > 0.0 and ms20 > 0.0 and ms200 > 3.0 * ms20, (ms1, ms20, ms200)  # replays cost time in proportion
        ids, votes = eng.batch_fetch(slot, det.n)
        ref = O.associate(cfg, tr, 1, det, want_matrices=False)
        np.testing.assert_array_equal(ids, ref["track_id"])
        np.testing.assert_array_equal(votes, ref["voting_type"])
    finally:
        eng.close()


@pytest.mark.paths("signal_completion")   # (the completion SIGNAL of the last dispatch instead of the scenes' completion words)
@pytest.mark.parametrize("poll_spin_us", [0, -1, 1], ids=["poll_default", "block_at_once", "poll_1us_then_block"])
@pytest.mark.parametrize("pinned", [False, True], ids=["pageable", "pinned_block"])
def test_pipelined_tickets_match_the_synchronous_path(pinned, poll_spin_us):
    """A stream of different frames through sa_pipe_submit / sa_pipe_wait with two (then three) tickets in flight: every frame's answer equals
    the oracle's (and therefore sa_associate's); features from pageable memory or DMA'd in place from sa_host_alloc blocks.  The wait for a
    set's completion words polls for sa_config.poll_spin_us and then blocks on the stream: the default, blocking at once (-1), and a 1 us
    budget that runs out on nearly every frame."""
    rng = np.random.default_rng(73)
    d, n, t = 128, 150, 170
    cfg = visual_cfg(d, poll_spin_us=poll_spin_us)
    sc = synth.visual_scene(rng, t, n, d, 1, canvas=(1500.0, 900.0), new_fraction=0.1)
    eng = Engine(cfg)
    blocks = []
    try:
        tr = upsert_scene(eng, 3, sc, True)
        frames = []
        for f in range(6):
            p = rng.permutation(n)[: n - 7 * f]  # ragged: every frame another size
            feats = sc["det_feats"][p]
            if pinned:
                blk = eng.host_block(feats.shape)
                blk[...] = feats
                blocks.append(blk)
                feats = blk
            det = abi.make_detections(sc["det_boxes"][p], feats=feats, feat_quality=sc["det_quality"][p])
            req, res, outs = Engine.make_requests([(3, 1, det)])
            frames.append((det, req, res, outs))
        tickets = []
        for f, (det, req, res, outs) in enumerate(frames):
            tickets.append(eng.pipe_submit(req))
            if f >= 1:
                eng.pipe_wait(tickets[f - 1], frames[f - 1][2])
        eng.pipe_wait(tickets[-1], frames[-1][2])
        for det, req, res, outs in frames:
            ref = O.associate(cfg, tr, 1, det, want_matrices=False)
            np.testing.assert_array_equal(outs[0][0], ref["track_id"])
            np.testing.assert_array_equal(outs[0][1], ref["voting_type"])
        # three tickets may be outstanding (one per bank); a fourth is refused, and so is a synchronous batch meanwhile
        t1 = eng.pipe_submit(frames[0][1])
        t2 = eng.pipe_submit(frames[1][1])
        t3 = eng.pipe_submit(frames[3][1])
        with pytest.raises(EngineError) as ei:
            eng.pipe_submit(frames[2][1])
        assert ei.value.code == abi.SA_ERR_STATE
        with pytest.raises(EngineError):
            eng.batch_begin()
        eng.pipe_wait(t2, frames[1][2])  # any order
        eng.pipe_wait(t1, frames[0][2])
        eng.pipe_wait(t3, frames[3][2])
        for k in (0, 1, 3):
            ref = O.associate(cfg, tr, 1, frames[k][0], want_matrices=False)
            np.testing.assert_array_equal(frames[k][3][0][0], ref["track_id"])
        with pytest.raises(EngineError):
            eng.pipe_wait(12345, frames[0][2])
        # the synchronous path still works afterwards and agrees
        ids, votes = eng.associate(3, 1, frames[2][0])
        np.testing.assert_array_equal(ids, frames[2][3][0][0])
    finally:
        for b in blocks:
            eng.host_free(b)
        eng.close()


def test_completion_words_never_arrive_before_the_results_they_announce():
    """A long pipelined run (three tickets in flight, request sets of one to four scenes of every size from 3 to ~400 detections, each
    frame different): the host takes a ticket's results the moment its scenes' completion words show the launch's sequence number —
    every id and vote type must be what the synchronous path answers for the same request (computed beforehand on a second engine).
    A word that overtook its results would hand out the previous occupant of the result block."""
    rng = np.random.default_rng(79)
    cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
    scenes = {s: synth.sort_scene(rng, 420, 420, canvas=(1400.0, 900.0), pos_sigma=3.0) for s in range(4)}
    a, b = Engine(cfg), Engine(cfg)
    try:
        for eng in (a, b):
            for s, sc in scenes.items():
                upsert_scene(eng, s, sc, False)
        sets = []
        for f in range(2000):
            items = []
            for s in rng.permutation(4)[: 1 + f % 4]:
                sc = scenes[int(s)]
                n = int(rng.integers(3, len(sc["det_boxes"])))
                p = rng.permutation(len(sc["det_boxes"]))[:n]
                items.append((int(s), 1, abi.make_detections(sc["det_boxes"][p])))
            want = [b.associate(s, 1, det) for s, _, det in items]
            sets.append((items, Engine.make_requests(items), want))
        depth = 3
        tickets = [None] * len(sets)
        for f in range(len(sets) + depth - 1):
            if f < len(sets):
                tickets[f] = a.pipe_submit(sets[f][1][0])
            j = f - (depth - 1)
            if j >= 0:
                items, (req, res, outs), want = sets[j]
                a.pipe_wait(tickets[j], res)
                for k, ((ids, votes), (wi, wv)) in enumerate(zip(outs, want)):
                    np.testing.assert_array_equal(ids, wi, err_msg=f"request set {j}, scene {k}")
                    np.testing.assert_array_equal(votes, wv)
    finally:
        a.close()
        b.close()


def test_apply_in_two_halves_matches_the_single_call():
    """sa_tracks_apply_begin / _end (the tracker facade's form: its own bookkeeping runs between the two) on oriented boxes against
    sa_tracks_apply on a second engine, six frames: same predicted boxes, same tables and polygons.  On odd frames the second half is
    left out until AFTER an entry point that needs the finished table (the polygon tap): it finishes what is pending, and _end
    afterwards only hands out the boxes.  _end without a _begin is refused."""
    rng = np.random.default_rng(75)
    cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
    world = synth.dense_boxes(rng, 50, (900.0, 700.0), True)
    frames = []
    for f in range(6):
        world = synth.jitter_boxes(rng, world, 1.5, angle_sigma=0.02)
        frames.append(np.concatenate([world, synth.dense_boxes(rng, 3, (900.0, 700.0), True)]))
        world = frames[-1]
    a, b = Engine(cfg), Engine(cfg)
    u64p, bp = C.POINTER(C.c_uint64), C.POINTER(abi.sa_box)
    try:
        pred0 = np.zeros(1, abi.BOX_DTYPE)
        nxt = 1
        for f, boxes in enumerate(frames):
            det = abi.make_detections(boxes)
            got = []
            for eng in (a, b):
                eng.batch_begin()
                eng.batch_add(0, f + 1, det)
                eng.batch_run()
                eng.batch_sync()
                got.append(eng.batch_fetch(0, det.n)[0])
            np.testing.assert_array_equal(got[0], got[1], err_msg=f"frame {f}")
            nid = np.zeros(det.n, np.uint64)
            for i, w in enumerate(got[0]):
                if w == 0:
                    nid[i] = nxt
                    nxt += 1
            if f == 0:
                assert b.lib.sa_tracks_apply_end(b.h, 0, C.cast(pred0.ctypes.data, bp)) == abi.SA_ERR_STATE
            pa, pb = np.zeros(det.n, abi.BOX_DTYPE), np.zeros(det.n, abi.BOX_DTYPE)
            a._chk(a.lib.sa_tracks_apply(a.h, 0, nid.ctypes.data_as(u64p), C.cast(pa.ctypes.data, bp)))
            b._chk(b.lib.sa_tracks_apply_begin(b.h, 0, nid.ctypes.data_as(u64p)))
            if f % 2:
                np.testing.assert_array_equal(b.tap_track_polygons(0), a.tap_track_polygons(0))  # finishes the pending half
            b._chk(b.lib.sa_tracks_apply_end(b.h, 0, C.cast(pb.ctypes.data, bp)))
            np.testing.assert_array_equal(pa.view(np.uint8), pb.view(np.uint8), err_msg=f"frame {f}")
            np.testing.assert_array_equal(a.order(0), b.order(0))
            np.testing.assert_array_equal(b.tap_track_polygons(0), a.tap_track_polygons(0))
        assert a.count(0) == b.count(0) > 50
    finally:
        a.close()
        b.close()


@pytest.mark.paths("signal_completion")   # (the completion SIGNAL of the last dispatch instead of the scenes' completion words)
def test_pipelined_tracker_loop_with_device_upkeep():
    """stage(n+1); wait(n); apply(n); launch(n+1): the H2D of the next frame overlaps the current frame's kernels, and the track
    table the next frame meets is the one sa_tracks_apply left (new tracks appended, Kalman steps taken).  Same ids, frame by
    frame, as the plain synchronous loop on a second engine."""
    rng = np.random.default_rng(74)
    cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
    n = 60
    world = synth.dense_boxes(rng, n, (900.0, 700.0))
    frames = []
    for f in range(6):
        world = synth.jitter_boxes(rng, world, 1.5)
        extra = synth.dense_boxes(rng, 3, (900.0, 700.0))  # a few objects appear every frame
        frames.append(np.concatenate([world, extra]))
        world = frames[-1]
    a, b = Engine(cfg), Engine(cfg)
    try:
        next_id = [1, 1]

        def new_ids(k, ids):
            out = np.zeros(len(ids), np.uint64)
            for i, w in enumerate(ids):
                if w == 0:
                    out[i] = next_id[k]
                    next_id[k] += 1
            return out

        def apply(eng, k, ids):
            nid = new_ids(k, ids)
            pred = np.zeros(len(ids), abi.BOX_DTYPE)
            eng._chk(eng.lib.sa_tracks_apply(eng.h, 0, nid.ctypes.data_as(C.POINTER(C.c_uint64)), C.cast(pred.ctypes.data, C.POINTER(abi.sa_box))))
            return pred

        # reference loop: synchronous
        sync_ids, sync_pred = [], []
        for f, boxes in enumerate(frames):
            det = abi.make_detections(boxes)
            a.batch_begin()
            a.batch_add(0, f + 1, det)
            a.batch_run()
            a.batch_sync()
            ids, _ = a.batch_fetch(0, det.n)
            sync_ids.append(ids)
            sync_pred.append(apply(a, 0, ids))
        # pipelined loop
        sets = [Engine.make_requests([(0, f + 1, abi.make_detections(boxes))]) for f, boxes in enumerate(frames)]
        tk = b.pipe_stage(sets[0][0])
        b.pipe_launch(tk)
        for f in range(len(frames)):
            nxt = b.pipe_stage(sets[f + 1][0]) if f + 1 < len(frames) else None
            b.pipe_wait(tk, sets[f][1])
            ids = sets[f][2][0][0]
            np.testing.assert_array_equal(ids, sync_ids[f], err_msg=f"frame {f}")
            pred = apply(b, 1, ids)
            np.testing.assert_array_equal(pred.view(np.uint8), sync_pred[f].view(np.uint8))
            if nxt is not None:
                b.pipe_launch(nxt)
                tk = nxt
        assert b.count(0) == a.count(0) > n
        assert (sync_ids[-1] != 0).sum() >= n  # continuing tracks are re-found on the table the device maintains
    finally:
        a.close()
        b.close()


def test_oriented_boxes_between_apply_begin_and_end_meet_the_next_launch_with_their_polygons():
    """stage(n+1); wait(n); apply_begin(n); launch(n+1); apply_end(n) with ORIENTED boxes: the upkeep kernel writes the axis-aligned
    polygon for refreshed oriented rows and the right one is queued only when the step is finished — sa_pipe_launch finishes a pending
    step before it queues kernels that read the table, so every frame's ids and the table's f64 polygons equal the synchronous loop's."""
    rng = np.random.default_rng(91)
    cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
    n = 70
    world = synth.dense_boxes(rng, n, (700.0, 500.0), oriented=True)
    world["angle"] = rng.uniform(-3.0, 3.0, n).astype(np.float32)
    frames = []
    for f in range(6):
        world = synth.jitter_boxes(rng, world, 1.5, angle_sigma=0.02)
        frames.append(world.copy())
    a, b = Engine(cfg), Engine(cfg)
    u64p, boxp = C.POINTER(C.c_uint64), C.POINTER(abi.sa_box)
    try:
        next_id = [1, 1]

        def new_ids(k, ids):
            out = np.zeros(len(ids), np.uint64)
            for i, w in enumerate(ids):
                if w == 0:
                    out[i] = next_id[k]
                    next_id[k] += 1
            return out

        sync_ids, sync_poly = [], []
        for f, boxes in enumerate(frames):
            det = abi.make_detections(boxes)
            a.batch_begin()
            a.batch_add(0, f + 1, det)
            a.batch_run()
            a.batch_sync()
            ids, _ = a.batch_fetch(0, det.n)
            sync_ids.append(ids)
            nid = new_ids(0, ids)
            pred = np.zeros(len(ids), abi.BOX_DTYPE)
            a._chk(a.lib.sa_tracks_apply(a.h, 0, nid.ctypes.data_as(u64p), C.cast(pred.ctypes.data, boxp)))
            sync_poly.append(a.tap_track_polygons(0))
        sets = [Engine.make_requests([(0, f + 1, abi.make_detections(boxes))]) for f, boxes in enumerate(frames)]
        tk = b.pipe_stage(sets[0][0])
        b.pipe_launch(tk)
        for f in range(len(frames)):
            nxt = b.pipe_stage(sets[f + 1][0]) if f + 1 < len(frames) else None
            b.pipe_wait(tk, sets[f][1])
            ids = sets[f][2][0][0]
            np.testing.assert_array_equal(ids, sync_ids[f], err_msg=f"frame {f}")
            nid = new_ids(1, ids)
            b._chk(b.lib.sa_tracks_apply_begin(b.h, 0, nid.ctypes.data_as(u64p)))
            if nxt is not None:
                b.pipe_launch(nxt)   # reads the table: must see the oriented polygons of the step queued above
            pred = np.zeros(len(ids), abi.BOX_DTYPE)
            b._chk(b.lib.sa_tracks_apply_end(b.h, 0, C.cast(pred.ctypes.data, boxp)))
            if nxt is not None:
                tk = nxt
            else:
                np.testing.assert_array_equal(b.tap_track_polygons(0).view(np.uint64), sync_poly[f].view(np.uint64))
        assert (sync_ids[-1] != 0).sum() >= n - 2
    finally:
        a.close()
        b.close()


@pytest.mark.paths("signal_completion")   # (the completion SIGNAL of the last dispatch instead of the scenes' completion words)
@pytest.mark.parametrize("per_candidate", [0, 1])
@pytest.mark.parametrize("visual", [False, True])
def test_upkeep_queued_behind_the_association_equals_the_two_phase_upkeep(visual, per_candidate):
    """sa_batch_run_apply / sa_tracks_apply_collect (the facade's path: the Kalman / feature-bank step queued right behind the assignment
    tail, the ids of new tracks drawn ON THE DEVICE from a counter in candidate order) against sa_associate + sa_tracks_apply with the
    ids drawn on the host: same winners, same new ids, same predicted boxes, same tables (f64 polygons of oriented boxes included),
    frame after frame — new objects every frame, two scenes in one request set."""
    rng = np.random.default_rng(97 + per_candidate + 2 * visual)
    d = 64
    cfg = (visual_cfg(d, k=2, visual_minimal_quality_collect=0.4) if visual else abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5))
    u64p, boxp = C.POINTER(C.c_uint64), C.POINTER(abi.sa_box)
    scenes = (4, 9)
    world = {sc: synth.dense_boxes(rng, 40, (800.0, 600.0), oriented=(sc == 9)) for sc in scenes}
    ident = {sc: synth.reid_identities(rng, 200, d) for sc in scenes}
    a, b = Engine(cfg), Engine(cfg)
    try:
        counter = [0, 0]
        for f in range(6):
            items = []
            for sc in scenes:
                world[sc] = np.concatenate([synth.jitter_boxes(rng, world[sc], 1.5, angle_sigma=0.02 if sc == 9 else 0.0),
                                            synth.dense_boxes(rng, 3, (800.0, 600.0), oriented=(sc == 9))])
                n = len(world[sc])
                kw = dict(feats=synth.observe(rng, ident[sc][:n]), feat_quality=rng.uniform(0.2, 1.0, n).astype(np.float32)) if visual else {}
                items.append((sc, f + 1, abi.make_detections(world[sc], **kw)))
            # two-phase reference on engine a: associate, draw ids on the host in the reference's order, sa_tracks_apply
            a.batch_begin()
            slots = [a.batch_add(sc, ep, det) for sc, ep, det in items]
            a.batch_run()
            a.batch_sync()
            ref = []
            base_a = []
            for (sc, ep, det), sl in zip(items, slots):
                base_a.append(counter[0])
                ids, votes = a.batch_fetch(sl, det.n)
                nid = np.zeros(det.n, np.uint64)
                for i in range(det.n):
                    if per_candidate:
                        counter[0] += 1
                    if ids[i] == 0:
                        if not per_candidate:
                            counter[0] += 1
                        nid[i] = counter[0]
                ref.append((ids, votes, nid))
            preds_a = []
            for sl, (ids, votes, nid) in zip(slots, ref):
                pred = np.zeros(len(ids), abi.BOX_DTYPE)
                a._chk(a.lib.sa_tracks_apply(a.h, sl, nid.ctypes.data_as(u64p), C.cast(pred.ctypes.data, boxp)))
                preds_a.append(pred)
            # fused on engine b
            if not per_candidate:
                # (a counter per NEW track across scenes would make a scene's first id depend on the earlier scenes' winners: one scene
                # per request set under that rule — what Sort / VisualSort::predict are)
                got = []
                for k, (sc, ep, det) in enumerate(items):
                    b.batch_begin()
                    sl = b.batch_add(sc, ep, det)
                    base = np.array([counter[1]], np.uint64)
                    b._chk(b.lib.sa_batch_run_apply(b.h, base.ctypes.data_as(u64p), 0))
                    b.batch_sync()
                    ids, votes = b.batch_fetch(sl, det.n)
                    nid = np.zeros(det.n, np.uint64)
                    pred = np.zeros(det.n, abi.BOX_DTYPE)
                    b._chk(b.lib.sa_tracks_apply_collect(b.h, sl, nid.ctypes.data_as(u64p), C.cast(pred.ctypes.data, boxp)))
                    counter[1] += int((ids == 0).sum())
                    got.append((ids, votes, nid, pred))
            else:
                b.batch_begin()
                slots_b = [b.batch_add(sc, ep, det) for sc, ep, det in items]
                bases, nxt = [], counter[1]
                for sc, ep, det in items:
                    bases.append(nxt)
                    nxt += det.n
                ba = np.array(bases, np.uint64)
                b._chk(b.lib.sa_batch_run_apply(b.h, ba.ctypes.data_as(u64p), 1))
                b.batch_sync()
                got = []
                for sl, (sc, ep, det) in zip(slots_b, items):
                    ids, votes = b.batch_fetch(sl, det.n)
                    nid = np.zeros(det.n, np.uint64)
                    pred = np.zeros(det.n, abi.BOX_DTYPE)
                    b._chk(b.lib.sa_tracks_apply_collect(b.h, sl, nid.ctypes.data_as(u64p), C.cast(pred.ctypes.data, boxp)))
                    got.append((ids, votes, nid, pred))
                counter[1] = nxt
            for k, ((ids, votes, nid), pred_a, (ids_b, votes_b, nid_b, pred_b)) in enumerate(zip(ref, preds_a, got)):
                np.testing.assert_array_equal(ids_b, ids, err_msg=f"frame {f} scene {k}")
                np.testing.assert_array_equal(votes_b, votes)
                np.testing.assert_array_equal(nid_b, nid, err_msg=f"frame {f} scene {k}: new ids")
                np.testing.assert_array_equal(pred_b.view(np.uint8), pred_a.view(np.uint8))
            for sc in scenes:
                assert a.count(sc) == b.count(sc)
                np.testing.assert_array_equal(b.tap_track_polygons(sc).view(np.uint64), a.tap_track_polygons(sc).view(np.uint64))
            if f == 0:
                assert (ref[0][0] == 0).all()   # first frame: every candidate starts a track
        assert a.count(4) > 40 and (ref[0][0] != 0).sum() >= (25 if visual else 40)   # tracks are continued, not only started
        # collecting twice is refused, not repeated
        with pytest.raises(EngineError):
            b._chk(b.lib.sa_tracks_apply_collect(b.h, 0, None, None))
    finally:
        a.close()
        b.close()


def test_graph_replay_survives_a_changing_epoch():
    """SA_FLAG_GRAPH under a tracker: the epoch (and the detections) change every frame, the launch geometry does not — the
    captured graph is replayed, and every frame's answer is the oracle's for THAT epoch (idle tracks drop out as it advances)."""
    rng = np.random.default_rng(75)
    cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=2, flags=abi.SA_FLAG_GRAPH)
    sc = synth.sort_scene(rng, 150, 150, canvas=(1200.0, 900.0))
    sc["track_epochs"] = rng.integers(0, 4, 150).astype(np.uint64)
    eng = Engine(cfg)
    try:
        tr = upsert_scene(eng, 0, sc, False)
        matched = []
        for epoch in (1, 2, 3, 4, 5):
            det = abi.make_detections(synth.jitter_boxes(rng, sc["det_boxes"], 0.5))
            ids, votes = eng.associate(0, epoch, det)
            ref = O.associate(cfg, tr, epoch, det, want_matrices=False)
            np.testing.assert_array_equal(ids, ref["track_id"], err_msg=f"epoch {epoch}")
            matched.append(int((ids != 0).sum()))
        assert matched[0] > matched[-1] > 0, matched  # compatible() really followed the epoch
    finally:
        eng.close()


def test_detection_features_read_in_place_from_device_memory():
    """sa_device_block_register: the candidates' feature rows are already in HBM (a ReID model on the same GPU wrote them) — the
    engine reads them where they lie.  Same answers as from host memory and as the oracle, through sa_associate, through the
    pipelined tickets, from a base that is not 16-byte aligned (one device-to-device copy), with ragged frames out of one block;
    a block registered for another device is refused.  Runs tests/devfeat_child.py: the stand-in for the ReID model's output is a
    torch tensor, and torch's HIP context wants to be the first one of its process."""
    import os
    import subprocess
    import sys

    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "devfeat_child.py")
    r = subprocess.run([sys.executable, child], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DEVICE-FEATURES-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_a_recycled_ticket_is_refused_not_misread():
    """sa_pipe_wait binds the slot numbers of sa_tracks_apply / sa_batch_fetch / the taps to the waited ticket's scenes.  Staging more
    request sets afterwards may recycle that ticket's bank: the engine then refuses those calls (SA_ERR_STATE) instead of applying the
    caller's old winners to the new set's scenes; and while another bank is idle the bound one is left alone."""
    rng = np.random.default_rng(75)
    cfg = abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5)
    eng = Engine(cfg)
    try:
        scs = [synth.sort_scene(rng, 60, 50 + 5 * i, canvas=(900.0, 700.0)) for i in range(5)]
        for i, sc in enumerate(scs):
            upsert_scene(eng, 20 + i, sc, False)
        sets = [Engine.make_requests([(20 + i, 1, abi.make_detections(sc["det_boxes"]))]) for i, sc in enumerate(scs)]
        t1 = eng.pipe_submit(sets[0][0])
        t2 = eng.pipe_submit(sets[1][0])
        eng.pipe_wait(t1, sets[0][1])
        ids1 = sets[0][2][0][0].copy()
        # one more set: a bank other than ticket 1's is idle, so slot 0 still means scene 20
        t3 = eng.pipe_submit(sets[2][0])
        np.testing.assert_array_equal(eng.batch_fetch(0, len(ids1))[0], ids1)
        assert eng.tap_dims(0)[0] == len(ids1)
        # a fourth set has to take ticket 1's bank: from now on its slots are gone
        t4 = eng.pipe_submit(sets[3][0])
        for call in (lambda: eng.batch_fetch(0, len(ids1)), lambda: eng.tap_dims(0), lambda: eng.tap_positional(0)):
            with pytest.raises(EngineError) as ei:
                call()
            assert ei.value.code == abi.SA_ERR_STATE
        new_ids = np.arange(1000, 1000 + len(ids1), dtype=np.uint64)
        pred = np.zeros(len(ids1), abi.BOX_DTYPE)
        rc = eng.lib.sa_tracks_apply(eng.h, 0, new_ids.ctypes.data_as(C.POINTER(C.c_uint64)), C.cast(pred.ctypes.data, C.POINTER(abi.sa_box)))
        assert rc == abi.SA_ERR_STATE and b"recycled" in eng.lib.sa_last_error(eng.h)
        # the tickets themselves are fine, and waiting rebinds
        for t, k in ((t2, 1), (t3, 2), (t4, 3)):
            eng.pipe_wait(t, sets[k][1])
            ref = O.associate(cfg, abi.make_tracks(scs[k]["track_ids"], scs[k]["track_boxes"], scs[k]["track_epochs"]), 1,
                              abi.make_detections(scs[k]["det_boxes"]), want_matrices=False)
            np.testing.assert_array_equal(sets[k][2][0][0], ref["track_id"])
        assert eng.tap_dims(0)[0] == len(scs[3]["det_boxes"])
    finally:
        eng.close()


@pytest.mark.paths("signal_completion")   # (the completion SIGNAL of the last dispatch instead of the scenes' completion words)
@pytest.mark.parametrize("visual", [False, True])
def test_request_set_staged_collected_and_evicted_on_several_threads(visual):
    """The entry points Batch*::predict spreads over threads, straight through the C ABI: sa_batch_add_deferred + sa_batch_fill (slots
    filled from a thread pool), sa_batch_results (zero-copy views), sa_tracks_apply_collect_begin / _slot / _end (slots collected from a
    thread pool), sa_tracks_remove_stage / _commit and sa_tracks_remove_many (several scenes' tables compacted by ONE launch) — against
    a second engine driven through the serial entry points (sa_batch_add, sa_batch_fetch, sa_tracks_apply_collect, sa_tracks_remove):
    same winners, ids, predicted boxes and tables, frame after frame."""
    from concurrent.futures import ThreadPoolExecutor

    rng = np.random.default_rng(131 + visual)
    d, S = 64, 14                                            # (14 scenes: the gathers of one removal go out as two launches of 12 + 2)
    cfg = (visual_cfg(d, k=2, visual_minimal_quality_collect=0.4) if visual else abi.make_config(positional="iou", positional_threshold=0.3, max_idle_epochs=5))
    u64p, u8p, i32p, boxp = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(abi.sa_box)
    scenes = [3 + 5 * s for s in range(S)]
    world = {sc: synth.dense_boxes(rng, 30 + sc % 7, (800.0, 600.0), oriented=(sc % 4 == 0)) for sc in scenes}
    ident = {sc: synth.reid_identities(rng, 120, d) for sc in scenes}
    a, b = Engine(cfg), Engine(cfg)
    pool = ThreadPoolExecutor(max_workers=6)
    try:
        nxt_id = [0, 0]
        for f in range(5):
            items = []
            for sc in scenes:
                world[sc] = np.concatenate([synth.jitter_boxes(rng, world[sc], 1.5, angle_sigma=0.02 if sc % 4 == 0 else 0.0),
                                            synth.dense_boxes(rng, 2, (800.0, 600.0), oriented=(sc % 4 == 0))])
                n = len(world[sc])
                kw = dict(feats=synth.observe(rng, ident[sc][:n]), feat_quality=rng.uniform(0.2, 1.0, n).astype(np.float32)) if visual else {}
                items.append((sc, f + 1, abi.make_detections(world[sc], **kw)))
            bases = []
            for e_i in (0, 1):
                acc, bs = nxt_id[e_i], []
                for sc, ep, det in items:
                    bs.append(acc)
                    acc += det.n
                bases.append(np.array(bs, np.uint64))
            # serial reference on engine a
            a.batch_begin()
            slots_a = [a.batch_add(sc, ep, det) for sc, ep, det in items]
            a._chk(a.lib.sa_batch_run_apply(a.h, bases[0].ctypes.data_as(u64p), 1))
            a.batch_sync()
            ref = []
            for sl, (sc, ep, det) in zip(slots_a, items):
                ids, votes = a.batch_fetch(sl, det.n)
                cols = np.zeros(det.n, np.int32)
                a._chk(a.lib.sa_batch_fetch_cols(a.h, sl, cols.ctypes.data_as(i32p)))
                nid, pred = np.zeros(det.n, np.uint64), np.zeros(det.n, abi.BOX_DTYPE)
                a._chk(a.lib.sa_tracks_apply_collect(a.h, sl, nid.ctypes.data_as(u64p), C.cast(pred.ctypes.data, boxp)))
                ref.append((ids, votes, cols, nid, pred))
            # engine b: deferred adds, fills / collects on the pool
            b.batch_begin()
            slots_b = []
            for sc, ep, det in items:
                sl = C.c_uint32()
                b._chk(b.lib.sa_batch_add_deferred(b.h, sc, ep, C.byref(det), None, C.byref(sl)))
                slots_b.append(sl.value)
            if f == 1:   # a slot that was never filled is refused by the run, and the set is still usable once it is
                rc = b.lib.sa_batch_run_apply(b.h, bases[1].ctypes.data_as(u64p), 1)
                assert rc == abi.SA_ERR_STATE and b"sa_batch_fill" in b.lib.sa_last_error(b.h)
            assert all(rc == 0 for rc in pool.map(lambda sl: b.lib.sa_batch_fill(b.h, sl), slots_b))
            b._chk(b.lib.sa_batch_run_apply(b.h, bases[1].ctypes.data_as(u64p), 1))

            def view(sl_n):
                sl, n = sl_n
                pi, pv, pc = u64p(), u8p(), i32p()
                rc = b.lib.sa_batch_results(b.h, sl, C.byref(pi), C.byref(pv), C.byref(pc))
                assert rc == 0, b.lib.sa_last_error(b.h)
                return (np.ctypeslib.as_array(pi, (n,)).copy(), np.ctypeslib.as_array(pv, (n,)).copy(), np.ctypeslib.as_array(pc, (n,)).copy()) if n else \
                       (np.zeros(0, np.uint64), np.zeros(0, np.uint8), np.zeros(0, np.int32))
            views = list(pool.map(view, [(sl, det.n) for sl, (_, _, det) in zip(slots_b, items)]))
            # the table side of every OTHER slot ahead of the wait for the Kalman dispatch (sa_tracks_apply_collect_table: it needs the winners
            # only; the new ids go out there, _slot then hands out the boxes alone), the rest in one step behind it
            early = {}

            def table(sl_n):
                sl, n = sl_n
                nid = np.zeros(n, np.uint64)
                rc = b.lib.sa_tracks_apply_collect_table(b.h, sl, nid.ctypes.data_as(u64p))
                assert rc == 0, b.lib.sa_last_error(b.h)
                assert b.lib.sa_tracks_apply_collect_table(b.h, sl, None) == 0      # (a second call finds it done)
                return sl, nid
            for sl, nid in pool.map(table, [(sl, det.n) for k, (sl, (_, _, det)) in enumerate(zip(slots_b, items)) if (k + f) % 2 == 0]):
                early[sl] = nid
            b._chk(b.lib.sa_tracks_apply_collect_begin(b.h))

            def collect(sl_n):
                sl, n = sl_n
                nid, pred = np.zeros(n, np.uint64), np.zeros(n, abi.BOX_DTYPE)
                rc = b.lib.sa_tracks_apply_collect_slot(b.h, sl, nid.ctypes.data_as(u64p), C.cast(pred.ctypes.data, boxp))
                assert rc == 0, b.lib.sa_last_error(b.h)
                return (early[sl] if sl in early else nid), pred
            got = list(pool.map(collect, [(sl, det.n) for sl, (_, _, det) in zip(slots_b, items)]))
            b._chk(b.lib.sa_tracks_apply_collect_end(b.h))
            for k, ((ids, votes, cols, nid, pred), (vi, vv, vc), (nid_b, pred_b)) in enumerate(zip(ref, views, got)):
                np.testing.assert_array_equal(vi, ids, err_msg=f"frame {f} slot {k}")
                np.testing.assert_array_equal(vv, votes)
                np.testing.assert_array_equal(vc, cols)
                np.testing.assert_array_equal(nid_b, nid)
                np.testing.assert_array_equal(pred_b.view(np.uint8), pred.view(np.uint8))
            for e_i in (0, 1):
                nxt_id[e_i] += sum(det.n for _, _, det in items)
            # every other frame: the tracks that did not continue leave the tables of ALL scenes — one call on b (staged on the pool where
            # the scene allows it), one call per scene on a
            if f % 2 == 1:
                lists = []
                for sc in scenes:
                    order = a.order(sc)
                    gone = order[: max(1, len(order) // 4)]
                    lists.append(np.ascontiguousarray(gone, np.uint64))
                    a._chk(a.lib.sa_tracks_remove(a.h, sc, len(gone), gone.ctypes.data_as(u64p)))
                if f == 1:
                    b.batch_sync()   # (a drained engine: every scene can be staged off the calling thread)
                    rcs = list(pool.map(lambda k: b.lib.sa_tracks_remove_stage(b.h, scenes[k], len(lists[k]), lists[k].ctypes.data_as(u64p)), range(S)))
                    assert all(rc == abi.SA_OK for rc in rcs), rcs
                    rest = []
                else:
                    rest = list(range(S))
                if rest:
                    sid = np.array([scenes[k] for k in rest], np.uint64)
                    cnt = np.array([len(lists[k]) for k in rest], np.uint32)
                    ptrs = (u64p * len(rest))(*[lists[k].ctypes.data_as(u64p) for k in rest])
                    b._chk(b.lib.sa_tracks_remove_many(b.h, len(rest), sid.ctypes.data_as(u64p), cnt.ctypes.data_as(C.POINTER(C.c_uint32)), ptrs))
                else:
                    b._chk(b.lib.sa_tracks_remove_commit(b.h))
            for sc in scenes:
                np.testing.assert_array_equal(b.order(sc), a.order(sc), err_msg=f"frame {f} scene {sc}")
                np.testing.assert_array_equal(b.tap_track_polygons(sc).view(np.uint64), a.tap_track_polygons(sc).view(np.uint64))
        # a scene listed twice, an unknown id: refused, and nothing has changed
        two = np.array([scenes[0], scenes[0]], np.uint64)
        one = np.ascontiguousarray(a.order(scenes[0])[:1], np.uint64)
        ptr2 = (u64p * 2)(one.ctypes.data_as(u64p), one.ctypes.data_as(u64p))
        assert b.lib.sa_tracks_remove_many(b.h, 2, two.ctypes.data_as(u64p), np.array([1, 1], np.uint32).ctypes.data_as(C.POINTER(C.c_uint32)), ptr2) == abi.SA_ERR_BAD_ARG
        bad = np.array([2 ** 60], np.uint64)
        assert b.lib.sa_tracks_remove_stage(b.h, scenes[1], 1, bad.ctypes.data_as(u64p)) == abi.SA_ERR_NOT_FOUND
        b._chk(b.lib.sa_tracks_remove_commit(b.h))
        for sc in scenes:
            np.testing.assert_array_equal(b.order(sc), a.order(sc))
    finally:
        pool.shutdown(wait=True)
        a.close()
        b.close()
