"""ctypes binding of include/sactd3.h.  Fails loudly when the HIP library is missing: there is no
CPU fallback for the product path."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ABI_VERSION = 1
ACTOR, CRITICS, ACTOR_TARGET, CRITICS_TARGET, LOG_ALPHA = range(5)
SITE_CRITIC, SITE_ACTOR0, SITE_ACTOR1, SITE_ALPHA0, SITE_ALPHA1, SITE_PREDICT = range(6)
NUM_METRICS = 8

# every symbol include/sactd3.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "sactd3_abi_version", "sactd3_default_config", "sactd3_create", "sactd3_destroy", "sactd3_last_error",
    "sactd3_param_count", "sactd3_get_params", "sactd3_set_params", "sactd3_get_adam_state", "sactd3_set_adam_state",
    "sactd3_rb_extend", "sactd3_rb_len", "sactd3_rb_sample", "sactd3_rb_sample_with_indices", "sactd3_load_batch",
    "sactd3_read_batch", "sactd3_rb_fill_synthetic", "sactd3_set_noise", "sactd3_clear_noise", "sactd3_read_noise",
    "sactd3_update_qnets", "sactd3_update_actor", "sactd3_update_targ_nets", "sactd3_step", "sactd3_predict",
    "sactd3_read_metrics", "sactd3_sync", "sactd3_debug_read", "sactd3_debug_names", "sactd3_graph_kernel_count",
    "sactd3_time_kernel", "sactd3_time_gather_sweep", "sactd3_time_nodes", "sactd3_rb_layout", "sactd3_rb_extend_device", "sactd3_step_period", "sactd3_step_prefix", "sactd3_instantiate_graphs", "sactd3_device_handles",
]


class EngineError(RuntimeError):
    pass


class CConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "ob_dim", "ac_dim", "batch_size", "rb_capacity", "max_envs", "prefer_td3_over_sac",
        "layer_norm", "autotune", "bcq_style_targ_mix", "targ_actor_smoothing", "actor_update_delay",
        "crit_targ_update_freq", "use_graphs", "device_id", "reserved0")] + [(n, C.c_float) for n in (
        "actor_lr", "qnets_lr", "log_alpha_lr", "gamma", "polyak", "alpha_init", "clip_norm", "td3_std", "td3_c",
        "actor_noise_std", "adam_beta1", "adam_beta2", "adam_eps", "reserved1")] + [("seed", C.c_uint64)]


def library_path() -> str:
    """In-tree build product; SACTD3_LIBRARY overrides it (A/B-testing kernel variants)."""
    return os.environ.get("SACTD3_LIBRARY") or os.path.join(_HERE, "libsactd3_hip.so")


def build_library(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> libsactd3_hip.so in-tree (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc")] + (["-B"] if force else [])
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return library_path()


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise EngineError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C sac-td3-cudagraphs-pytorch_amd/csrc`). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.  If our library (linked against
    # /opt/rocm's) were loaded first, a later `import torch` would bring a second runtime that finds no GPU.  Importing
    # torch first makes the dynamic linker resolve our dependency to the copy already loaded.  (No torch: nothing to do.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    fp, u8p, i64p, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int64), C.c_void_p
    sig = {
        "sactd3_abi_version": (C.c_int, []),
        "sactd3_default_config": (None, [C.POINTER(CConfig), C.c_int]),
        "sactd3_create": (C.c_int, [C.POINTER(CConfig), fp, fp, C.POINTER(vp)]),
        "sactd3_destroy": (None, [vp]),
        "sactd3_last_error": (C.c_char_p, [vp]),
        "sactd3_param_count": (C.c_int64, [vp, C.c_int]),
        "sactd3_get_params": (C.c_int, [vp, C.c_int, fp]),
        "sactd3_set_params": (C.c_int, [vp, C.c_int, fp]),
        "sactd3_get_adam_state": (C.c_int, [vp, C.c_int, fp, fp, i64p]),
        "sactd3_set_adam_state": (C.c_int, [vp, C.c_int, fp, fp, C.c_int64]),
        "sactd3_rb_extend": (C.c_int, [vp, fp, fp, fp, fp, u8p, C.c_int]),
        "sactd3_rb_len": (C.c_int64, [vp]),
        "sactd3_rb_layout": (C.c_int, [vp, C.POINTER(C.c_int32)]),
        "sactd3_rb_extend_device": (C.c_int, [vp, C.c_void_p, C.c_int]),
        "sactd3_rb_sample": (C.c_int, [vp]),
        "sactd3_rb_sample_with_indices": (C.c_int, [vp, i64p, C.c_int]),
        "sactd3_load_batch": (C.c_int, [vp, fp, fp, fp, fp, u8p, C.c_int]),
        "sactd3_read_batch": (C.c_int, [vp, fp, fp, fp, fp, u8p, i64p]),
        "sactd3_rb_fill_synthetic": (C.c_int, [vp, C.c_int64, C.c_uint64]),
        "sactd3_set_noise": (C.c_int, [vp, C.c_int, fp, C.c_int]),
        "sactd3_clear_noise": (C.c_int, [vp, C.c_int]),
        "sactd3_read_noise": (C.c_int, [vp, C.c_int, fp, C.c_int]),
        "sactd3_update_qnets": (C.c_int, [vp]),
        "sactd3_update_actor": (C.c_int, [vp]),
        "sactd3_update_targ_nets": (C.c_int, [vp, C.c_int64]),
        "sactd3_step": (C.c_int, [vp, C.c_int]),
        "sactd3_step_period": (C.c_int, [vp]),
        "sactd3_step_prefix": (C.c_int, [vp, C.c_int]),
        "sactd3_instantiate_graphs": (C.c_int, [vp]),
        "sactd3_predict": (C.c_int, [vp, fp, C.c_int, C.c_int, fp]),
        "sactd3_read_metrics": (C.c_int, [vp, fp]),
        "sactd3_sync": (C.c_int, [vp]),
        "sactd3_device_handles": (C.c_int, [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
        "sactd3_debug_read": (C.c_int64, [vp, C.c_char_p, fp, C.c_int64]),
        "sactd3_debug_names": (C.c_char_p, []),
        "sactd3_graph_kernel_count": (C.c_int, [vp, C.c_int]),
        "sactd3_time_kernel": (C.c_int, [vp, C.c_char_p, C.c_int, fp]),
        "sactd3_time_gather_sweep": (C.c_int, [vp, C.c_int, C.c_int, fp, C.POINTER(C.c_double)]),
        "sactd3_time_nodes": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, fp, C.POINTER(C.c_double), C.POINTER(C.c_double), i64p]),
    }
    assert sorted(sig) == sorted(SYMBOLS)
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # AttributeError here = the .so does not export what the header declares
        fn.restype, fn.argtypes = res, args
    if lib.sactd3_abi_version() != ABI_VERSION:
        raise EngineError("libsactd3_hip.so ABI version mismatch; rebuild it")
    _LIB = lib
    return lib
