import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["SACTD3_LIBRARY"] = "/root/repo/sac-td3-cudagraphs-pytorch_amd/libsactd3_hip_tune.so"
import test_gpu_engine as T
from test_gpu_engine import *
_lib = T._lib
def run(rows4, B=96, env="humanoid"):
    os.environ["SACTD3_ROWS4"] = str(rows4)
    ref, eng, (o, a, bound) = make_pair("sac", env, B, True)
    obs, act, rew, nobs, done = synth_transitions(B, o, a, bound, seed=3)
    done[::5] = True
    eps = torch.randn(B, a, generator=torch.Generator().manual_seed(4))
    eng.load_batch(obs, act, rew, nobs, done)
    eng.set_noise(_lib.SITE_CRITIC, eps)
    eng.update_qnets()
    return {k: eng.debug_read(k).copy() for k in ("c_dz1", "c_dh1", "grad_critics", "c_dz2")}, (o, a)
a, (o, ac) = run(4096); b, _ = run(0); c, _ = run(0)
for k in a:
    print(k, "fold vs nofold", np.abs(a[k] - b[k]).max(), "fold vs fold", np.abs(b[k] - c[k]).max(), "max", np.abs(a[k]).max())
ga, gb = a["grad_critics"].reshape(2, -1), b["grad_critics"].reshape(2, -1)
d = np.abs(ga - gb)
for n in range(2):
    idx = np.argsort(-d[n])[:12]
    print(n, idx, d[n][idx])
