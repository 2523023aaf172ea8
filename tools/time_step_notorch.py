"""steps/s of the fused iteration in a process that never imports torch (HIP runtime = /opt/rocm's, not torch's bundled one)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "torch-first":
    import torch
import sac_td3_cudagraphs_pytorch_amd as P
from sac_td3_cudagraphs_pytorch_amd import _lib
eng = P.Engine(P.Config(ob_dim=11, ac_dim=3, batch_size=256, rb_capacity=1000000), [-1] * 3, [1] * 3)
rng = np.random.default_rng(0)
for which in (0, 2): eng.set_params(which, (rng.standard_normal(eng.param_count(0)) * 0.05).astype(np.float32))
for which in (1, 3): eng.set_params(which, (rng.standard_normal(eng.param_count(1)) * 0.05).astype(np.float32))
eng.rb_fill_synthetic(100000, 0); eng.sync()
for i in range(300): eng.step(i % 3 == 0)
eng.sync()
t = time.perf_counter()
for i in range(3000): eng.step(i % 3 == 0)
eng.sync()
dt = time.perf_counter() - t
print(sys.argv[1:] or "no torch", "steps/s %.0f  us/step %.1f" % (3000 / dt, dt / 3000 * 1e6), eng.read_metrics())
import ctypes
print([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:1])
rows = [np.zeros((4, 11), np.float32), np.zeros((4, 3), np.float32), np.zeros(4, np.float32), np.zeros((4, 11), np.float32), np.zeros(4, bool)]
for _ in range(40): eng.rb_extend(*rows)
eng.sync(); t = time.perf_counter()
for _ in range(300): eng.rb_extend(*rows)
eng.sync(); print("rb_extend %.1f us" % ((time.perf_counter() - t) / 300 * 1e6))
for _ in range(20): eng.predict(rows[0], True)
t = time.perf_counter()
for _ in range(300): eng.predict(rows[0], True)
print("predict %.1f us" % ((time.perf_counter() - t) / 300 * 1e6))
