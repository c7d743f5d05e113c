"""Do a frame's kernels run slower when the device has been idle since the previous frame?  C3's 8-scene request set (resident inputs):
phase A = 60 x (sa_batch_run; sync) back to back, phase B = the same with the host busy-waiting `gap_us` between frames, phase C = A
again.  Run under `rocprofv3 --kernel-trace`; scripts/gpu_idle_gap.sh splits the trace by phase (the phases are separated by a 20 ms sleep).
   python scripts/idle_gap_probe.py [workload] [gap_us]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from similari_amd.engine import Engine  # noqa: E402

wname = sys.argv[1] if len(sys.argv) > 1 else "c3"
gap = float(sys.argv[2]) * 1e-6 if len(sys.argv) > 2 else 60e-6
cfg, scenes, _ = bench.workload(wname, seed=1234)[:3]
eng = Engine(cfg)
keep, dets = bench.stage(eng, cfg, scenes)
for _ in range(10):
    eng.batch_run()
eng.batch_sync()


def phase(n, wait):
    t0 = time.perf_counter()
    for _ in range(n):
        eng.batch_run()
        eng.batch_sync()
        if wait:
            t1 = time.perf_counter() + wait
            while time.perf_counter() < t1:
                pass
    return (time.perf_counter() - t0) / n


time.sleep(0.02)
a = phase(60, 0.0)
time.sleep(0.02)
b = phase(60, gap)
time.sleep(0.02)
c = phase(60, 0.0)
print({"workload": wname, "gap_us": gap * 1e6, "us_per_frame_back_to_back": round(a * 1e6, 1), "us_per_frame_with_gaps": round(b * 1e6 - gap * 1e6, 1), "again_back_to_back": round(c * 1e6, 1)})
eng.close()
