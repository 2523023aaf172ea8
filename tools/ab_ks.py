import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for ks in ("", "1", "2", "4"):
    env = dict(os.environ)
    if ks: env["SACTD3_KS"] = ks
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-baselines", "--steps", "2000"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print("KS=%s" % (ks or "auto"), round(d["value"]), round(d["ms_per_step"] * 1e3, 1), "trunk4 %.2f us" % d["roofline_mfma"]["avg_launch_us"], flush=True)
