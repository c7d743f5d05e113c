"""Child process of tests/test_gpu_pipeline.py::test_detection_features_read_in_place_from_device_memory — a process of its own
because it needs torch's HIP context (the stand-in for a ReID model's output buffer) next to the engine's, initialised first."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.zeros(1, device="cuda:0")  # torch's context first, as in a process that runs its detector before the tracker

import oracle_lib as O  # noqa: E402
from similari_amd import abi, synth  # noqa: E402
from similari_amd.engine import Engine, EngineError  # noqa: E402


def visual_cfg(d, k=1, **kw):
    return abi.make_config(positional="iou", positional_threshold=0.3, visual="cosine", visual_threshold=0.2, feature_len=d,
                           max_observations=k, visual_min_votes=1, visual_minimal_track_length=1, positional_min_confidence=0.1,
                           max_idle_epochs=5, **kw)


def main():
    rng = np.random.default_rng(76)
    d, n, t = 128, 150, 170
    cfg = visual_cfg(d)
    sc = synth.visual_scene(rng, t, n, d, 1, canvas=(1500.0, 900.0), new_fraction=0.1)
    eng = Engine(cfg)
    try:
        tr = abi.make_tracks(sc["track_ids"], sc["track_boxes"], sc["track_epochs"], feats=sc["track_feats"], feat_present=sc["track_present"])
        eng.upsert(5, tr)
        pool = torch.zeros(4 * n * d + 8, dtype=torch.float32, device="cuda:0")  # the "ReID output buffer"
        eng.register_device_block(pool.data_ptr(), pool.numel() * 4, 0)
        frames = []
        off = 0
        for f in range(4):
            p = rng.permutation(n)[: n - 11 * f]
            feats = np.ascontiguousarray(sc["det_feats"][p])
            start = off + (1 if f == 2 else 0)  # frame 2 starts 4 bytes off a 16-byte boundary
            pool[start:start + feats.size] = torch.from_numpy(feats.ravel()).to("cuda:0")
            ptr = pool.data_ptr() + 4 * start
            off = (start + feats.size + 3) // 4 * 4
            host = abi.make_detections(sc["det_boxes"][p], feats=feats, feat_quality=sc["det_quality"][p])
            dev = abi.make_detections(sc["det_boxes"][p], feats_device_ptr=ptr, feat_quality=sc["det_quality"][p])
            frames.append((host, dev))
        torch.cuda.synchronize()
        assert pool.data_ptr() % 16 == 0
        for f, (host, dev) in enumerate(frames):
            ref = O.associate(cfg, tr, 1, host, want_matrices=False)
            ids_h, votes_h = eng.associate(5, 1, host)
            ids_d, votes_d = eng.associate(5, 1, dev)
            np.testing.assert_array_equal(ids_d, ref["track_id"], err_msg=f"frame {f}")
            np.testing.assert_array_equal(votes_d, ref["voting_type"])
            np.testing.assert_array_equal(ids_d, ids_h)
            assert (votes_d == abi.SA_VOTE_VISUAL).sum() > 0.5 * len(ids_d)  # the features really were read
        # the same frames as pipelined tickets, three in flight
        sets = [Engine.make_requests([(5, 1, dev)]) for _, dev in frames]
        tks = [eng.pipe_submit(sets[f][0]) for f in range(3)]
        for f in range(3):
            eng.pipe_wait(tks[f], sets[f][1])
        tk = eng.pipe_submit(sets[3][0])
        eng.pipe_wait(tk, sets[3][1])
        for f, (host, dev) in enumerate(frames):
            ref = O.associate(cfg, tr, 1, host, want_matrices=False)
            np.testing.assert_array_equal(sets[f][2][0][0], ref["track_id"], err_msg=f"ticket {f}")
        # a block registered for another device is refused; afterwards the engine is still usable
        eng.unregister_device_block(pool.data_ptr())
        eng.register_device_block(pool.data_ptr(), pool.numel() * 4, 1)
        try:
            eng.associate(5, 1, frames[0][1])
            raise AssertionError("a block of device 1 was accepted by an engine on device 0")
        except EngineError as ex:
            assert ex.code == abi.SA_ERR_BAD_ARG and "device" in str(ex), str(ex)
        eng.unregister_device_block(pool.data_ptr())
        ids_h, _ = eng.associate(5, 1, frames[0][0])
        np.testing.assert_array_equal(ids_h, O.associate(cfg, tr, 1, frames[0][0], want_matrices=False)["track_id"])
    finally:
        eng.close()
    print("DEVICE-FEATURES-OK")


if __name__ == "__main__":
    main()
