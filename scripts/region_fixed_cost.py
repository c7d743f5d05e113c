"""What a K-step timed region of bench.py carries besides its K steps (measurement only): intercept of region time against K, and the
cost of the brackets on an idle device."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from similari_amd.engine import Engine

torch.cuda.set_device(0)
cfg, scenes, desc = bench.workload("c2", seed=1234)
cfg.device = 0
cfg.flags = bench.DEFAULT_FLAGS
eng = Engine(cfg)
keep = bench.stage(eng, cfg, scenes, list(range(len(scenes))))
for _ in range(50): eng.batch_run()
eng.batch_sync()
pc = time.perf_counter
def med(f, n=200):
    xs = []
    for _ in range(n):
        t0 = pc(); f(); xs.append(pc() - t0)
    return np.median(xs) * 1e6
print("torch.cuda.synchronize() idle: %.2f us" % med(torch.cuda.synchronize))
print("eng.batch_sync() idle: %.2f us" % med(eng.batch_sync))
def region(k, end):
    def f():
        for _ in range(k): eng.batch_run()
        end()
    return f
def end_a(): eng.batch_sync(); torch.cuda.synchronize(); torch.cuda.synchronize()
def end_b(): eng.batch_sync(); torch.cuda.synchronize()
def end_c(): torch.cuda.synchronize()
def end_d(): eng.batch_sync()
for name, end in (("batch_sync only", end_d), ("batch_sync + 2 synchronize (bench.py)", end_a), ("batch_sync + synchronize", end_b), ("synchronize", end_c)):
    ks = [5, 10, 20, 40, 80, 160]
    ts = []
    for k in ks:
        torch.cuda.synchronize(); torch.cuda.synchronize()
        ts.append(med(region(k, end), 60))
    slope, icpt = np.polyfit(ks, ts, 1)
    print(f"{name}: " + " ".join(f"K={k}:{t / k:.2f}" for k, t in zip(ks, ts)) + f" | slope {slope:.2f} us/step, intercept {icpt:.1f} us")
# host cost of one batch_run (no GPU wait)
t0 = pc()
for _ in range(200): eng.batch_run()
t1 = pc(); eng.batch_sync()
print("host time per batch_run while the queue fills: %.2f us" % ((t1 - t0) / 200 * 1e6))
