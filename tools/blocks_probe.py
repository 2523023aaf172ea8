#!/usr/bin/env python3
"""Per-block timeline of one launch inside a critic-only iteration, from a stamps build (make -C sac-td3-cudagraphs-pytorch_amd/csrc
stamps): the critics' weight-gradient launch (k_tn, default) or the 4-net critic trunk (k_nt): python tools/blocks_probe.py [workload] [tn|nt]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SACTD3_LIBRARY"] = os.path.join(ROOT, "sac-td3-cudagraphs-pytorch_amd", "libsactd3_hip_stamps.so")
import numpy as np
import bench
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "hopper_sac"]
which = 1 if len(sys.argv) > 2 and sys.argv[2] in ("nt", "ntp", "ntr") else 0
rider = len(sys.argv) > 2 and sys.argv[2] == "ntr"      # a cut-short period of 2 iterations: its last trunk launch carries the deferred temperature step
period = len(sys.argv) > 2 and sys.argv[2] == "ntp"     # a whole period: the run-ahead trunk launch survives in the block ids the critic trunks do not reach
names = (["descriptor + decode", "operands landed", "mfma done", "reduced", "committed"],
         ["descriptor + decode", "operands parked", "layer 1 + LN", "layer-1 stores", "layer 2", "reduced", "stored"])[which]
eng = bench.make_engine(w, 0, 0)
it = bench.run_steps(eng, 0, 30); eng.sync()
lib = eng.lib
lib.sactd3_debug_blocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
lib.sactd3_debug_phases.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
lib.sactd3_debug_blocks_select.argtypes = [C.c_void_p, C.c_int]
n = 1600
for rep in range(3):
    assert lib.sactd3_debug_blocks_select(eng._h, which) == 0
    if rider:
        eng.step_period(); eng.sync(); assert lib.sactd3_debug_blocks_select(eng._h, which) == 0; eng.step_prefix(2); eng.sync()
    elif period:
        eng.step_period(); eng.sync()
    else:
        eng.step(False); eng.step(False); eng.sync()      # the last stamped launch of the family = the second iteration's
    buf = (C.c_longlong * (2 * n))()
    assert lib.sactd3_debug_blocks(eng._h, buf, n) == 0
    a = np.array(buf[:], np.int64).reshape(n, 2)
    live = np.where(a[:, 0] > 0)[0]
    t0 = a[live, 0].min()
    beg, end = (a[:, 0] - t0) * 10, (a[:, 1] - t0) * 10   # ns
    print(f"rep {rep}: {len(live)} blocks; launch span {end[live].max()} ns; begin: p50 {np.median(beg[live]):.0f} max {beg[live].max()} ns; duration p50 {np.median((end - beg)[live]):.0f} max {(end - beg)[live].max()} ns")
    if rep == 2:
        ph = (C.c_longlong * (8 * n))()
        assert lib.sactd3_debug_phases(eng._h, ph, n) == 0
        P = (np.array(ph[:], np.int64).reshape(n, 8) - t0) * 10
        raw = np.array(ph[:], np.int64).reshape(n, 8)
        cyc = (raw[live, 7] - raw[live, 6]).astype(np.float64); ns = (end - beg)[live].astype(np.float64)
        ok = (cyc > 0) & (ns > 0)
        print(f"shader clock while these blocks ran: median {np.median(cyc[ok] / ns[ok]):.3f} GHz")
        P[:, 6:] = -10**9
        print("phases:", names)
        for b in live:
            k = int((P[b] > -1000).sum())
            print(f"  block {b:4d} begin {beg[b]:6d} " + " ".join(f"{P[b, i]:6d}" for i in range(k)) + f" | end {end[b]:6d} dur {end[b] - beg[b]:6d}")
