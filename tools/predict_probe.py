#!/usr/bin/env python3
"""Acting-side latency probe: python tools/predict_probe.py [workload] -- predict / rb_extend / sync call times (host clock)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "hopper_sac"]
eng = bench.make_engine(w, 0, 0)
ob = np.zeros((4, w["o"]), np.float32)
def timeit(f, n=2000, warm=200):
    for _ in range(warm): f()
    t = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t) / n * 1e6
print("predict explore   us", round(timeit(lambda: eng.predict(ob, True)), 2))
print("predict exploit   us", round(timeit(lambda: eng.predict(ob, False)), 2))
print("sync (idle)       us", round(timeit(eng.sync), 2))
print("predict 1 row     us", round(timeit(lambda: eng.predict(ob[:1], True)), 2))
eng.close()
