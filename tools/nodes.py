#!/usr/bin/env python3
"""Per-node device times of one workload's iteration graphs (kernel tuning aid): python tools/nodes.py humanoid_sac [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "hopper_sac"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
w = bench.WORKLOADS[name]
eng = bench.make_engine(w, 0, 0)
it = bench.run_steps(eng, 0, 300); eng.sync()
import time
t0 = time.perf_counter(); bench.run_steps(eng, it, 3000); eng.sync(); dt = time.perf_counter() - t0
print(f"{name}: {3000/dt:.0f} it/s  {dt/3000*1e6:.2f} us/iteration")
eng.close()
nu, rl, _ = bench.node_budget(w, name, 0, dt / 3000 * 1e3, iters)
for k in ("sum_critic_only_us", "sum_critic_plus_2_actor_us", "sum_period_us", "sum_per_iteration_us", "unaccounted_us_per_iteration"):
    print(k, round(nu[k], 2))
for n, us in nu["period_of_3_iterations"]:
    print(f"  {us:7.2f}  {n}")
for r in rl[:8]:
    print(f"{r['kernel']:28s} grids {r['grids']} {r['avg_launch_us']:6.2f} us x{r['launches_per_iteration']:.2f}/it  {r['bound']} frac {r['frac']:.3f}  share {r['share_of_node_time']:.3f}")
