"""Soak run (design aid): N fused iterations with rb_extend / predict traffic in between, metrics checked for finiteness.
python tools/soak.py [workload] [iterations]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "hopper_sac"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
w = bench.WORKLOADS[name]
eng = bench.make_engine(w, 0, 0)
rows = [np.random.randn(4, w["o"]).astype(np.float32), np.zeros((4, w["a"]), np.float32), np.ones(4, np.float32),
        np.random.randn(4, w["o"]).astype(np.float32), np.zeros(4, bool)]
ob = np.random.randn(4, w["o"]).astype(np.float32)
t0 = time.time()
for i in range(n):
    if i % 8 == 0:
        eng.rb_extend(*rows)
    if i % 64 == 0:
        act = eng.predict(ob, True)
        assert np.isfinite(act).all() and np.abs(act).max() <= w["bound"] * 1.5 + 1e-6, (i, act)
    eng.step(i % 3 == 0)
    if i % 20000 == 0:
        m = eng.read_metrics()
        assert all(np.isfinite(v) for v in m.values()), (i, m)
        print(i, {k: round(v, 4) for k, v in m.items()}, "%.0f it/s" % (i / max(time.time() - t0, 1e-9)), flush=True)
m = eng.read_metrics()
p = eng.get_params(0)
assert all(np.isfinite(v) for v in m.values()) and np.isfinite(p).all()
print("soak ok", n, m, "rb_len", eng.rb_len())
