"""bench.py's result line: the driver parses the LAST stdout line out of an 8 KB tail (round 2's 23.5 KB line came back
`parsed: null`), so the line is built by `bench.result_line` from the full record and must stay under bench.LINE_MAX bytes
with every key of the contract on it; the full record goes to the side file."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")
ROOFLINE = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algo_flops_per_launch", "avg_launch_us")
CPU = ("value", "unit", "cores", "kind", "sample")


def _roofline(kernel):
    return {"kernel": kernel, "grids": [65536, 131072], "roles": ["x" * 80] * 6, "bound": "mfma", "achieved": 19.181419609231238,
            "peak": 157.3, "unit": "TFLOP/s", "frac": 0.12194163769377772, "traffic": 4236163.0, "traffic_source": "profiles/r03_pmc_hopper_sac.csv",
            "algo_flops_per_launch": 141557760.0, "algo_bytes_per_launch": 3248128.0, "avg_launch_us": 7.3799417813618975,
            "launches_per_iteration": 1.0, "us_per_iteration": 7.3799417813618975, "share_of_node_time": 0.11007817752830512}


def _full_record(w):
    """the shape of what bench.main() + extras() assemble, with long-winded contents everywhere the real one has them"""
    nodes = [[f"k_nt<1,true,4,1,1>:critic/next-action+sample & actor0/policy/layers1+2 #{i}", 6.629] for i in range(28)]
    sec = {"workload": "y" * 240, "value": 6847.3123456, "unit": "gradient-steps/s", "steps": 3000, "warmup": 300, "ms_per_step": 0.146041234,
           "value_windows": {"median": 6847.3, "min": 6800.1, "max": 6900.2, "repeats": 5, "steps": 3000},
           "node_us": {"critic_only": nodes[:10], "critic_plus_2_actor": nodes}, "roofline": _roofline("k_nt64<2,2,1>"),
           "rooflines_top": [_roofline("k_nt64<4,2,2>")] * 4, "rooflines_per_grid": [_roofline("k_nn")] * 20,
           "parity": {"max_abs_delta": {"loss/qf_loss": 5.3e-5}, "max_rel": 9e-6}, "speedup_vs_eager_rocm": 23.6412345,
           "cpu_baseline": {"value": 72.3, "unit": "gradient-steps/s", "cores": 8, "kind": "port", "sample": "z" * 150}}
    return {
        "metric": "gradient-steps/sec (SAC, batch=256) at 1 GPU + 8-seed node; replay HBM GB/s", "value": 15292.84365704928,
        "unit": "gradient-steps/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.06539006233409357, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": bench.describe(w), "parallelism": "1 independent seeds, one per GPU, no collective"},
        "critic_updates_per_s": 15292.8, "actor_updates_per_s": 10195.2,
        "kernels_per_iteration": {"period_of_3_iterations": 41, "critic_only": 7, "critic_plus_2_actor": 28, "average_per_iteration": 13.666666666666666},
        "final_metrics": {"loss/qf_loss": 1.7554147243499756, "loss/actor_loss": -1.2096941471099854, "loss/alpha_loss": 0.195, "vitals/alpha": 0.0411},
        "value_3000_steps": {"median": 15285.69304199603, "min": 15100.123456789, "max": 15400.987654321, "repeats": 5, "steps": 3000},
        "replay_gather_cold": {"kernel": "k_gather", "batch": 256, "us_incl_counter_tick_kernel": 5.668, "algo_bytes": 54784, "GB/s": 9.66},
        "predict_round_trip_us": 15.5, "rb_extend_call_us": 8.17, "loop_with_acting_per_s": 10638.2,
        "learners_per_gpu_aggregate_steps_per_s": {"2": 23246.5, "4": 23459.9, "8": 28672.8},
        "gather_batch_sweep": {str(b): {"us": 5.0, "GB/s": 10.0} for b in (256, 4096, 65536)},
        "node_us": {"critic_only": nodes[:7], "critic_plus_2_actor": nodes, "note": "n" * 170},
        "roofline": _roofline("k_nn"), "rooflines_top": [_roofline("k_tn<2>")] * 4, "rooflines_per_grid": [_roofline("k_nn")] * 16,
        "replay_gather_hbm": {"record": "Humanoid-v4 (o=376, a=17), 1M-row ring", "rows_per_launch": 65536, "us": 85.25045013427734,
                              "algo_bytes": 404094976.0, "achieved": 4740.091992048288, "peak": 8000.0, "unit": "GB/s", "frac": 0.5925114990060361,
                              "bound": "hbm", "traffic": 419613120.0, "traffic_unit": "bytes per launch", "traffic_source": "profiles/x.csv"},
        "parity": {"path": "p" * 110, "iterations": 6, "max_abs_delta": {"loss/qf_loss": 1.4e-6}, "max_rel_delta": {"loss/qf_loss": 8e-7},
                   "max_rel": 8.123456e-7, "note": "q" * 170},
        "cpu_baseline": {"value": 183.79740832660494, "unit": "gradient-steps/s", "cores": 8, "kind": "port",
                         "sample": "1104 iterations of this workload, oracle/sac_td3_ref.py, PyTorch CPU eager, 8 threads (faster of 1 and 8)"},
        "eager_rocm_baseline": {"value": 315.9, "unit": "gradient-steps/s", "sample": "s" * 76}, "speedup_vs_eager_rocm": 48.404490206507646,
        "torch_cudagraph_rocm_baseline": {"value": 484.9, "unit": "gradient-steps/s", "sample": "t" * 170}, "speedup_vs_torch_cudagraph_rocm": 31.537236370899254,
        "configs": {"halfcheetah_td3": sec, "humanoid_sac": sec}, "detail": bench.DETAIL_FILE,
    }


def test_result_line_is_short_and_carries_the_contract():
    w = bench.WORKLOADS["hopper_sac"]
    full = _full_record(w)
    assert len(json.dumps(full)) > 20_000                       # the record itself is the size that broke round 2
    line = bench.result_line(full, w)
    assert "\n" not in line and len(line.encode()) < bench.LINE_MAX <= 1800
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    assert set(d["config"]) == {"workload", "parallelism"} and "model" not in d["config"]
    for k in ROOFLINE:
        assert k in d["roofline"], k
    assert d["roofline"]["kernel"] == "k_nn" and abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    for k in CPU:
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] == "port" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["value_3000_steps"]["repeats"] == 5 and d["value_3000_steps"]["min"] <= d["value_3000_steps"]["median"] <= d["value_3000_steps"]["max"]
    assert abs(d["value"] - full["value"]) / full["value"] < 1e-4 and abs(d["ms_per_step"] - full["ms_per_step"]) / full["ms_per_step"] < 1e-4
    assert d["parity_max_rel"] < 1e-5 and d["speedup_vs_eager_rocm"] > 5
    for name in ("halfcheetah_td3", "humanoid_sac"):
        assert {"value", "ms_per_step", "roofline_frac"} <= set(d["configs"][name])
    assert d["detail"] == bench.DETAIL_FILE


def test_result_line_of_a_timed_only_or_multi_rank_run():
    """N > 1 and --timed-only lines carry the timed region only: same contract keys, still one short line."""
    w = bench.WORKLOADS["hopper_sac"]
    full = {k: v for k, v in _full_record(w).items() if k in CONTRACT + ("kernels_per_iteration", "final_metrics")}
    full["n_gpus"] = 8
    d = json.loads(bench.result_line(full, w))
    for k in CONTRACT:
        assert k in d, k
    assert "roofline" not in d and "cpu_baseline" not in d


def test_roofline_groups_sum_the_grids_of_one_kernel_instance():
    """the dominant kernel is chosen by instance NAME (rocprofv3 --stats' grouping), its frac is work-weighted over its launches"""
    mk = lambda name, us, flops, by, th: dict(name=name, us=us, flops=flops, bytes=by, threads=th)
    g0 = [mk("k_nn:critic/dh1", 3.0, 33.5e6, 1.5e6, 4096), mk("k_nt<1,true,2,1,2>:critic/trunk", 7.0, 141.6e6, 3.2e6, 65536)]
    g1 = g0 + [mk("k_nn:actor/dh1", 3.5, 33.5e6, 1.5e6, 4096), mk("k_nn:actor/dq_da", 4.0, 8e6, 0.5e6, 1024), mk("k_nn:actor/dh1'", 3.5, 33.5e6, 1.5e6, 4096)]
    traffic = {("k_nn", 4096): dict(traffic=4e6, calls=30, source="f.csv"), ("k_nn", 1024): dict(traffic=1e6, calls=10, source="f.csv")}
    per_iter = (2 * sum(n["us"] for n in g0) + sum(n["us"] for n in g1)) / 3
    by_name, by_grid = bench.roofline_groups([(2.0 / 3.0, g0), (1.0 / 3.0, g1)], traffic, per_iter)
    nn = [r for r in by_name if r["kernel"] == "k_nn"][0]
    assert nn["grids"] == [1024, 4096] and abs(nn["launches_per_iteration"] - (1 + 3 / 3)) < 1e-9
    us = (2 * 3.0 + (3.0 + 3.5 + 4.0 + 3.5)) / 3
    fl = (2 * 33.5e6 + (3 * 33.5e6 + 8e6)) / 3
    assert abs(nn["us_per_iteration"] - us) < 1e-9 and abs(nn["achieved"] - fl / us * 1e-6) < 1e-9
    assert abs(nn["frac"] - nn["achieved"] / bench.MFMA_F32_PEAK_TF) < 1e-12 and abs(nn["traffic"] - (4e6 * 30 + 1e6 * 10) / 40) < 1e-6
    assert len([r for r in by_grid if r["kernel"] == "k_nn"]) == 2
    assert abs(sum(r["share_of_node_time"] for r in by_name) - 1.0) < 1e-9
