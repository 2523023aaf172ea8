#!/usr/bin/env python3
"""Fixed cost of a timed window: T(K periods) = a + K * period for K = 1, 2, 4, 8, 16 (median of 30 windows each), bracketed like
bench.py's timed region (engine sync + torch.cuda.synchronize): python tools/window_fit.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "hopper_sac"]
eng = bench.make_engine(w, 0, 0)
it = bench.run_steps(eng, 0, 300); eng.sync()
res = {}
for K in (1, 2, 4, 8, 16):
    ts = []
    for rep in range(30):
        eng.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        it = bench.run_steps(eng, it, 3 * K)
        t1 = time.perf_counter()
        eng.sync()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        ts.append(((t3 - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6))
    a = np.median(np.array(ts), axis=0)
    res[K] = a[0]
    print(f"K={K:2d} periods: total {a[0]:7.1f} us  (launch calls {a[1]:6.1f}, engine sync {a[2]:7.1f}, torch sync {a[3]:5.1f})")
Ks = np.array(sorted(res)); T = np.array([res[k] for k in Ks])
slope, icpt = np.polyfit(Ks, T, 1)
print(f"fit: {slope:.1f} us per period ({slope / 3:.2f} per iteration) + {icpt:.1f} us fixed")
