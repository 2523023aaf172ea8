import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for v in sys.argv[1:] or ("base", "cpt2", "cpt8"):
    env = dict(os.environ, SACTD3_LIBRARY=os.path.join(ROOT, "tools", "libs", f"lib_{v}.so"))
    res = []
    for b in (1024, 16384, 65536):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "humanoid_sac", "--gather-profile", str(b)],
                             env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out); res.append((b, round(d["us"], 1), round(d["GB/s"])))
    print(v, res, flush=True)
