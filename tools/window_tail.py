#!/usr/bin/env python3
"""What the left-over iterations of a window cost: windows of 18, 19, 20, 21 iterations starting at an iteration with actor updates
(median of 30 each, bracketed like bench.py's timed region): python tools/window_tail.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "hopper_sac"]
eng = bench.make_engine(w, 0, 0)
it = bench.run_steps(eng, 0, 300); eng.sync()
for n in (18, 19, 20, 21, 1, 2):
    ts = []
    for rep in range(30):
        it = bench.run_steps(eng, it, (-it) % 3 + 3)       # up to a period boundary, then one whole period: the chain is ready
        eng.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        it = bench.run_steps(eng, it, n)
        eng.sync(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    print(f"{n:2d} iterations: median {np.median(ts):7.1f} us  min {min(ts):7.1f}")
