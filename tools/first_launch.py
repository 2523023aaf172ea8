#!/usr/bin/env python3
"""First launch of an instantiated graph vs the following ones (wall time of launch + sync, us)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "hopper_sac"]
eng = bench.make_engine(w, 0, 0)
def timed(f):
    eng.sync(); t0 = time.perf_counter(); f(); eng.sync(); return (time.perf_counter() - t0) * 1e6
print("critic-only single :", [round(timed(lambda: eng.step(False)), 1) for _ in range(4)])
print("period (A,B,A,B)   :", [round(timed(lambda: eng.step_period()), 1) for _ in range(4)])
print("actor single       :", [round(timed(lambda: eng.step(True)), 1) for _ in range(4)])
print("critic-only single :", [round(timed(lambda: eng.step(False)), 1) for _ in range(2)])
print("period after single:", [round(timed(lambda: eng.step_period()), 1) for _ in range(3)])
