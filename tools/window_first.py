#!/usr/bin/env python3
"""The driver's window (--steps 20 --warmup 5) on a fresh engine, then the same window again and again: first-use effects.
python tools/window_first.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "hopper_sac"]
eng = bench.make_engine(w, 0, 0)
it = bench.run_steps(eng, 1, 5)
for rep in range(8):
    eng.sync(); torch.cuda.synchronize()
    stamps = []
    t0 = time.perf_counter()
    i, end = it, it + 20
    while i < end:
        if i % 3 == 0 and i + 3 <= end:
            eng.step_period(); i += 3
        else:
            eng.step(i % 3 == 0); i += 1
        stamps.append(round((time.perf_counter() - t0) * 1e6))
    it = i
    t1 = time.perf_counter()
    eng.sync()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"window {rep}: total {(t3 - t0) * 1e6:7.1f} us | launches returned at {stamps} | engine sync {1e6 * (t2 - t1):6.1f} torch sync {1e6 * (t3 - t2):5.1f}")
    it = bench.run_steps(eng, it, (-it) % 3 + 3)
