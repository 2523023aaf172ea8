"""Soak run of the chained period graphs: N iterations through run_iterations in chunks of mixed length (whole periods, cut-short
periods, singles), metrics and parameters checked for finiteness: python tools/soak_periods.py [workload] [iterations]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "hopper_sac"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600000
w = bench.WORKLOADS[name]
eng = bench.make_engine(w, 0, 0)
rng = np.random.default_rng(0)
it, t0, nxt = 0, time.time(), 0
while it < n:
    it = eng.run_iterations(it, int(rng.integers(1, 40)))
    if it >= nxt:
        m = eng.read_metrics()
        assert all(np.isfinite(v) for v in m.values()), (it, m)
        print(it, {k: round(v, 4) for k, v in m.items()}, "%.0f it/s" % (it / max(time.time() - t0, 1e-9)), flush=True)
        nxt += 100000
m = eng.read_metrics()
assert all(np.isfinite(v) for v in m.values()) and np.isfinite(eng.get_params(0)).all() and np.isfinite(eng.get_params(1)).all()
print("soak ok", it, m)
