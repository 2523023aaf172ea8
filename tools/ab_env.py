"""A/B an environment switch of the engine over bench.py: python tools/ab_env.py VAR v1 v2 ... [-- bench args]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
var, vals = args[0], args[1:]
for v in vals:
    env = dict(os.environ); env[var] = v
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-baselines", "--steps", "3000"] + extra, env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print("%s=%s" % (var, v), round(d["value"]), "%.2f us" % (d["ms_per_step"] * 1e3), d["kernels_per_iteration"], flush=True)
