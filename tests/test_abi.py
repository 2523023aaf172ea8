"""CPU: the C-ABI library loads without a GPU, exports every symbol include/sactd3.h declares, fails loudly
where a device is needed, and the ctypes mirror of `sactd3_config` matches the C layout."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "sactd3.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sactd3_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import sac_td3_cudagraphs_pytorch_amd as pkg
    from sac_td3_cudagraphs_pytorch_amd import _lib
    if not os.path.exists(pkg.library_path()):
        pkg.build_library()
    lib = C.CDLL(pkg.library_path())
    declared = header_symbols()
    assert declared == sorted(_lib.SYMBOLS)        # the binding covers the header, nothing more, nothing less
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/sactd3.h but not exported"
    assert pkg.load_library().sactd3_abi_version() == _lib.ABI_VERSION


def test_config_struct_layout_and_defaults():
    import sac_td3_cudagraphs_pytorch_amd as pkg
    from sac_td3_cudagraphs_pytorch_amd import _lib
    lib = pkg.load_library()
    assert C.sizeof(_lib.CConfig) == 128 and _lib.CConfig.seed.offset == 120
    for td3 in (0, 1):
        cc = _lib.CConfig()
        lib.sactd3_default_config(C.byref(cc), td3)
        want = pkg.Config(prefer_td3_over_sac=bool(td3), bcq_style_targ_mix=bool(td3), qnets_lr=3e-4 if td3 else 1e-3)
        for f, _ in _lib.CConfig._fields_:
            if f.startswith("reserved") or f in ("abi_version",):
                continue
            got, exp = getattr(cc, f), getattr(want, f)
            assert got == pytest.approx(float(exp), rel=1e-6), f   # tasks/defaults/{sac,td3}.yml values


def test_no_cpu_fallback():
    """Without a GPU the product path refuses to run instead of computing somewhere else."""
    import torch
    import sac_td3_cudagraphs_pytorch_amd as pkg
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.EngineError, match="no HIP device|not gfx950"):
        pkg.Engine(pkg.Config(ob_dim=11, ac_dim=3), -1.0, 1.0)
    assert pkg.load_library().sactd3_param_count(None, 0) < 0   # NULL engine -> error code, not a crash


def test_product_package_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing the product ships may import, include or execute it."""
    pkg_dir = os.path.join(ROOT, "sac-td3-cudagraphs-pytorch_amd")
    bad = re.compile(r"^\s*(from\s+oracle|import\s+oracle)|#\s*include\s*[\"<][^\">]*oracle|importlib[^\n]*oracle", re.M)
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert not bad.search(open(os.path.join(dirpath, f)).read()), f


def test_no_kernel_spills_to_scratch():
    """Scratch (private) memory costs a global-memory round trip per access; a dynamically indexed local array is
    enough to trigger it.  Every kernel must report ScratchSize 0 in hipcc's resource-usage remarks."""
    import subprocess
    csrc = os.path.join(ROOT, "sac-td3-cudagraphs-pytorch_amd", "csrc")
    out = subprocess.run(["make", "-C", csrc, "asm"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    names = re.findall(r"Function Name: (\S+)", out)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out)]
    assert len(names) == len(scratch) and len(names) >= 15, out[-2000:]
    bad = [(n, s) for n, s in zip(names, scratch) if s != 0]
    assert not bad, bad


def test_no_serialised_operand_loads():
    """A load first used inside a divergent branch, or issued after a possibly aliasing store, makes the compiler drain
    the memory queue (s_waitcnt vmcnt(0)) before the next request goes out: an operand fetch written as
    `k < K ? load : 0` waits a full round trip per load.  Check the compiled gfx950 code of every kernel: no run of
    load -> drain -> load -> drain, and the GEMM kernels issue their operand fetch as one batch."""
    import subprocess
    csrc = os.path.join(ROOT, "sac-td3-cudagraphs-pytorch_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "asm"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True)
    txt = open("/tmp/sactd3_engine.s").read()
    seqs = {}
    for name in re.findall(r"^(_Z[\w]+):\s*;?.*$", txt, re.M):
        i = txt.index("\n" + name + ":")
        ev = []
        for line in txt[i:txt.index(".Lfunc_end", i)].splitlines():
            ins = line.strip()
            if ins.startswith(("global_load", "buffer_load")):
                ev.append("L")
            elif ins.startswith(("global_store", "buffer_store")):
                ev.append("S")
            elif ins.startswith("s_waitcnt") and "vmcnt(0)" in ins:
                ev.append("W")
        seqs[name] = "".join(ev)
    assert len(seqs) >= 15
    for name, seq in seqs.items():
        if "k_rb_fill" in name:          # synthetic-data filler of the bench, not on any path
            continue
        assert "LWLWLW" not in seq, (name, seq)
    # (mangled names: "4k_tnI" = the k_tn<KT> instances, "6k_tn64I" = k_tn64<...>, whose prologue requests two chunks of 4 + 2 float4; "_Z4k_nn" = k_nn alone,
    # "6k_nn64I" = the LDS-tiled k_nn64<...> instances: 4 to 6 float4 per chunk)
    for key, batch in (("k_nt64", 4), ("k_nt_wide", 16), ("4k_tnI", 16), ("6k_tn64I", 12), ("_Z4k_nn", 16), ("6k_nn64I", 4), ("k_critic_tail", 32), ("k_actor_tail", 32)):
        hit = [s for n, s in seqs.items() if key in n]
        assert hit and all("L" * batch in s for s in hit), (key, hit)


def test_shipped_library_reads_no_environment_switches():
    """include/sactd3.h: every behaviour of libsactd3_hip.so is a field of sactd3_config; the kernel-selection A/B switches
    (SACTD3_KS, SACTD3_ROWS4, ...) and the any-arch override exist only in the tuning build (-DSACTD3_TUNING)."""
    import sac_td3_cudagraphs_pytorch_amd as pkg
    blob = open(os.path.join(os.path.dirname(pkg.__file__), "libsactd3_hip.so"), "rb").read()
    for name in (b"SACTD3_KS", b"SACTD3_NT", b"SACTD3_ROWS4", b"SACTD3_XR", b"SACTD3_TN64_MIN", b"SACTD3_TN_KT", b"SACTD3_PAD64", b"SACTD3_NN16",
                 b"SACTD3_ALLOW_ANY_ARCH"):
        assert name not in blob, name
    src = open(os.path.join(ROOT, "sac-td3-cudagraphs-pytorch_amd", "csrc", "engine.hip")).read()
    outside = re.sub(r"#ifdef SACTD3_TUNING.*?#endif", "", src, flags=re.S)
    assert "getenv" not in outside


def test_fetch_width_table_matches_the_code_objects():
    """tools/pmc_summary.py doubles rocprofv3's FETCH_SIZE (gfx950 tallies 16-byte-per-lane reads at half, MI355X_MICROARCH.md, HBM)
    only for kernels whose operand fetches are all global_load_dwordx4; its MIXED table names the kernels with 4-byte strided
    operand loads.  Check the table against the compiled gfx950 code: share of non-dwordx4 bytes among a kernel's load
    instructions (static count; the few dword loads of control words / biases / counters every kernel has stay below 0.3)."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary
    csrc = os.path.join(ROOT, "sac-td3-cudagraphs-pytorch_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "asm"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True)
    txt = open("/tmp/sactd3_engine.s").read()
    width = {"dwordx4": 16, "dwordx3": 12, "dwordx2": 8, "dword": 4, "ushort": 2, "sshort": 2, "ubyte": 1, "sbyte": 1}
    names = re.findall(r"^(_Z[\w]+):\s*;?.*$", txt, re.M)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.split("\n")
    seen = set()
    for name, d in zip(names, dem):
        i = txt.index("\n" + name + ":")
        wide = narrow = 0
        for line in txt[i:txt.index(".Lfunc_end", i)].splitlines():
            m = re.match(r"\s*(?:global|buffer)_load_(\w+)", line)
            if m:
                b = width[m.group(1)]
                wide, narrow = (wide + 16, narrow) if b == 16 else (wide, narrow + b)
        inst = pmc_summary.norm(d)
        if wide + narrow < 64 or not inst.startswith("k_"):     # counter / flag kernels: no operand fetch to speak of
            continue
        seen.add(inst)
        assert (narrow / (wide + narrow) >= 0.3) == (pmc_summary.fetch_factor(inst) == 1.0), (inst, wide, narrow)
    assert set(pmc_summary.MIXED) <= seen and len(seen) >= 40
