#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes into the per-kernel HBM-traffic summary that bench.py reads (profiles/rNN_pmc_<workload>.csv).

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python3 bench.py --timed-only ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 bench.py --timed-only ...
  python tools/pmc_summary.py --workload hopper_sac --fetch gpurun_out/pmc_f --write gpurun_out/pmc_w --out profiles/r02_pmc_hopper_sac.csv

Separate passes (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md, rocprofv3 PMC slots).  Units and
corrections as that guide's HBM section prescribes: both counters are in KB; on gfx950 FETCH_SIZE reports half of the bytes
of 16-byte-per-lane coalesced reads, so fetch_bytes = 2 x 1024 x FETCH_SIZE for the kernels whose global loads are ALL
dwordx4 (`fetch_factor` 2, column `loads` = "dwordx4"); a kernel that also fetches operands with 4-byte strided loads
(MIXED below: k_nn's weight columns, the wide head backward, the dQ/da slice product) is outside what the guide calibrates:
its FETCH_SIZE is taken as is (`fetch_factor` 1, `loads` = "mixed: uncalibrated"), which may understate it by up to 2x.
WRITE_SIZE is exact: write_bytes = 1024 x WRITE_SIZE.
Rows are grouped by (kernel instance, Grid_Size = total threads), which is how bench.py's node registry names a launch.
"""
import argparse
import csv
import glob
import os
import re
from collections import defaultdict


# kernels (by name prefix) with 4-byte strided global loads beside their dwordx4 ones; every other k_* kernel of csrc/kernels.h
# fetches with float4 (global_load_dwordx4) only -- tests/test_abi.py::test_fetch_width_table_matches_the_code_objects checks the ISA (share of non-dwordx4 load bytes)
MIXED = ("k_nn", "k_actor_head_bwd", "k_ln_bwd<16>", "k_qtail_nn<2>")   # exact instance names; k_qtail_nn<2>: W2 / W1 fragments and the dQ/da epilogue's layer-1 operands are dword loads (32 % of its load bytes)        # exact instance names (no template arguments = a plain kernel)


def fetch_factor(kernel):
    return 1.0 if kernel in MIXED else 2.0


def norm(name):
    name = re.sub(r"^void\s+", "", name.strip())
    name = re.sub(r"\([^()]*\)\s*$", "", name)
    return name.replace(" ", "")


def load(path, counter):
    files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                a = acc[(norm(r["Kernel_Name"]), int(r["Grid_Size"]))]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True)
    ap.add_argument("--fetch", required=True, help="output dir (or counter_collection.csv) of the --pmc FETCH_SIZE pass")
    ap.add_argument("--write", required=True, help="... of the --pmc WRITE_SIZE pass")
    ap.add_argument("--out", required=True)
    ap.add_argument("--min-calls", type=int, default=5)
    a = ap.parse_args()
    fe, wr = load(a.fetch, "FETCH_SIZE"), load(a.write, "WRITE_SIZE")
    with open(a.out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["workload", "kernel", "threads", "calls", "FETCH_SIZE_kb_avg", "WRITE_SIZE_kb_avg", "fetch_bytes", "write_bytes",
                    "traffic_bytes", "avg_us_profiled", "fetch_factor", "loads"])
        for key in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 0, 0])[2])):
            f, x = fe.get(key, [0, 0.0, 0.0]), wr.get(key, [0, 0.0, 0.0])
            calls = max(f[0], x[0])
            if calls < a.min_calls or not key[0].startswith("k_"):
                continue
            fk, wk = (f[1] / f[0] if f[0] else 0.0), (x[1] / x[0] if x[0] else 0.0)
            ff = fetch_factor(key[0])
            fb, wb = ff * 1024.0 * fk, 1024.0 * wk
            w.writerow([a.workload, key[0], key[1], calls, f"{fk:.3f}", f"{wk:.3f}", f"{fb:.0f}", f"{wb:.0f}", f"{fb + wb:.0f}",
                        f"{(f[2] / f[0] if f[0] else x[2] / max(x[0], 1)):.2f}", f"{ff:.0f}", "dwordx4" if ff == 2.0 else "mixed: uncalibrated"])
    print("wrote", a.out)


if __name__ == "__main__":
    main()
