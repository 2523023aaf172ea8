#!/usr/bin/env python3
"""Per-block timeline of the critics' weight-gradient launch (k_tn) inside a critic-only iteration, from a stamps build
(make -C sac-td3-cudagraphs-pytorch_amd/csrc stamps): python tools/blocks_probe.py [workload]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SACTD3_LIBRARY"] = os.path.join(ROOT, "sac-td3-cudagraphs-pytorch_amd", "libsactd3_hip_stamps.so")
import numpy as np
import bench
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "hopper_sac"]
eng = bench.make_engine(w, 0, 0)
it = bench.run_steps(eng, 0, 30); eng.sync()
lib = eng.lib
lib.sactd3_debug_blocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
n = 400
for rep in range(3):
    eng.step(False); eng.step(False); eng.sync()          # the last stamped launch = the second iteration's k_tn
    buf = (C.c_longlong * (2 * n))()
    assert lib.sactd3_debug_blocks(eng._h, buf, n) == 0
    a = np.array(buf[:], np.int64).reshape(n, 2)
    ok = a[:, 0] > 0
    t0 = a[ok, 0].min()
    beg, end = (a[:, 0] - t0) * 10, (a[:, 1] - t0) * 10   # ns
    live = np.where(ok)[0]
    print(f"rep {rep}: {len(live)} blocks; launch span {end[live].max()} ns; begin: p50 {np.median(beg[live]):.0f} max {beg[live].max()} ns; duration p50 {np.median((end - beg)[live]):.0f} max {(end - beg)[live].max()} ns")
    if rep == 2:
        ph = (C.c_longlong * (8 * n))()
        lib.sactd3_debug_phases.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
        assert lib.sactd3_debug_phases(eng._h, ph, n) == 0
        P = (np.array(ph[:], np.int64).reshape(n, 8)[:, :5] - t0) * 10
        for b in live:
            if P[b, 0] > -1000:
                print(f"  phases {b:4d} begin {beg[b]:5d} | loads issued {P[b,0]:5d} landed {P[b,1]:5d} mfma done {P[b,2]:5d} reduced {P[b,3]:5d} committed {P[b,4]:5d} | end {end[b]:5d}")
        for b in live:
            print(f"  block {b:4d} begin {beg[b]:6d} end {end[b]:6d} dur {end[b] - beg[b]:6d}")
