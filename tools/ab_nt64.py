import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for v in sys.argv[1:] or ("0", "1", "2", "3"):
    env = dict(os.environ); env["SACTD3_NT64"] = v
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "humanoid_sac", "--no-baselines", "--steps", "2000"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print("NT64=%s" % v, round(d["value"]), round(d["ms_per_step"] * 1e3, 1), "trunk4 %.2f us" % d["roofline_mfma"]["avg_launch_us"], flush=True)
