#!/usr/bin/env python3
"""Where a short timed window (the driver's `--steps 20 --warmup 5`) spends its time: host time of every launch call, the final
wait, and the same window repeated (first-launch effects): python tools/window_probe.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "hopper_sac"]
eng = bench.make_engine(w, 0, 0)
it = bench.run_steps(eng, 1, 5); eng.sync()
for rep in range(4):
    stamps = []
    t0 = time.perf_counter()
    i, end = it, it + 20
    while i < end:
        if i % 3 == 0 and i + 3 <= end:
            eng.step_period(); i += 3
        else:
            eng.step(i % 3 == 0); i += 1
        stamps.append(time.perf_counter() - t0)
    it = i
    t1 = time.perf_counter()
    eng.sync()
    t2 = time.perf_counter()
    print(f"window {rep}: launches returned at us {[round(x * 1e6) for x in stamps]}  last launch -> sync return {1e6 * (t2 - t1):.0f} us  total {1e6 * (t2 - t0):.0f} us = {1e6 * (t2 - t0) / 20:.2f} us/step")
    it += (-it) % 3
    it = bench.run_steps(eng, it, 0)
