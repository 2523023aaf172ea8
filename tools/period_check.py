import sys, time; sys.path.insert(0, '.')
import bench, numpy as np
for name in ("hopper_sac", "halfcheetah_td3", "humanoid_sac"):
    w = bench.WORKLOADS[name]
    res = []
    for mode in ("step", "period"):
        eng = bench.make_engine(w, 0, 0)
        it = 0
        run = (lambda i, n: bench.run_steps(eng, i, n)) if mode == "step" else (lambda i, n: eng.run_iterations(i, n))
        it = run(it, 300); eng.sync()
        t0 = time.perf_counter(); it = run(it, 3000); eng.sync(); dt = time.perf_counter() - t0
        res.append((dt / 3000 * 1e6, eng.get_params(1).copy(), eng.get_params(0).copy(), eng.graph_kernel_count(4)))
        eng.close()
    print(name, "step %.2f us  period %.2f us  nodes/period %d  bit-equal %s" % (res[0][0], res[1][0], res[1][3], np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])))
