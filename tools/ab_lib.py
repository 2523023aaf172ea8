"""A/B two builds of the library over bench.py: python tools/ab_lib.py libA.so libB.so [-- bench args] (alternating, 2 rounds)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
for rnd in range(2):
    for lib in args:
        env = dict(os.environ)
        if lib != "default": env["SACTD3_LIBRARY"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-baselines", "--steps", "3000"] + extra, env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        print(os.path.basename(lib), round(d["value"]), "%.2f us" % (d["ms_per_step"] * 1e3), d["kernels_per_iteration"], flush=True)
