"""CPU checks of the drop-in boundary: libw2a.so builds for gfx950, loads without a GPU, and
exports exactly the entry points include/w2a.h declares; host-side argument validation works
without touching a device."""
import ctypes as C
import os
import re

import pytest

from weather2alert_amd import _ffi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_lib()
    return _ffi.load()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "w2a.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(w2a_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    syms = header_symbols()
    assert syms == sorted(_ffi.SYMBOLS)
    for s in syms:
        assert hasattr(lib, s), s


def test_host_only_entry_points(lib):
    assert lib.w2a_abi_version() == _ffi.ABI_VERSION == 18
    assert lib.w2a_state_bytes(0) == 0
    n = 1000
    b = lib.w2a_state_bytes(n)
    assert b >= 256 + 56 * n and b % 256 == 0
    # NULL / bad arguments are rejected on the host with a message, nothing is launched
    h = C.c_void_p()
    assert lib.w2a_create(None, 10, 0, None, 0, None, C.byref(h)) == -1
    assert b"NULL" in lib.w2a_last_error()
    assert lib.w2a_step(None, None, 0, None, None, None, None, 0, None) == -1
    assert lib.w2a_reset_device_rng(None, 0, -1, 0, -1, 0, 1, 1, None, None, None) == -1
    assert lib.w2a_group_workspace_bytes(0, 10, 10) == 0 and lib.w2a_group_workspace_bytes(1000, 10, 10) >= 4 * 4 * 1000 + 10 * 10 * 512
    assert lib.w2a_group_by_column(None, None, 0, None) == -1
    assert lib.w2a_posterior_mean_reward(None, None, 0, None, None) == -1
    # round-3 entry points: run-time choice of the posterior-mean kernel, bookkeeping queries, invalidation
    assert lib.w2a_set_posterior_kernel(None, _ffi.PM_KERNELS["matrix_i8"]) == -1
    assert lib.w2a_query(None, _ffi.Q_LOCKSTEP_DAY) == -1 and lib.w2a_invalidate(None, None) == -1
    assert _ffi.PM_KERNELS == {"vector": 0, "matrix": 1, "matrix_i8": 2}


def test_header_enums_match_binding():
    """Numeric constants the Python binding hard-codes against the enums of include/w2a.h."""
    text = open(os.path.join(ROOT, "include", "w2a.h")).read()

    def enum(name):
        m = re.search(r"\b" + name + r"\s*=?\s*(-?\d+)", text)  # enumerator `NAME = n` or `#define NAME n`
        assert m, name
        return int(m.group(1))

    for name, val in (("W2A_STEP_AUTORESET", _ffi.STEP_AUTORESET), ("W2A_STEP_NO_OBS", _ffi.STEP_NO_OBS),
                      ("W2A_STEP_CLASSIC", _ffi.STEP_CLASSIC), ("W2A_STEP_REWARD_GIVEN", _ffi.STEP_REWARD_GIVEN),
                      ("W2A_STEP_WIDE", _ffi.STEP_WIDE), ("W2A_STEP_SKIP_FINISHED", _ffi.STEP_SKIP_FINISHED),
                      ("W2A_STEP_UNPACKED", _ffi.STEP_UNPACKED), ("W2A_STEP_NEXT_STEP", _ffi.STEP_NEXT_STEP), ("W2A_STEP_NO_CAPTURE", _ffi.STEP_NO_CAPTURE), ("W2A_PM_VECTOR", 0), ("W2A_PM_MATRIX_F64", 1),
                      ("W2A_PM_MATRIX_I8", 2), ("W2A_Q_LOCKSTEP_DAY", _ffi.Q_LOCKSTEP_DAY),
                      ("W2A_Q_PACKED_ELIGIBLE", _ffi.Q_PACKED_ELIGIBLE), ("W2A_Q_PACKED_CURRENT", _ffi.Q_PACKED_CURRENT),
                      ("W2A_Q_CANONICAL_CURRENT", _ffi.Q_CANONICAL_CURRENT), ("W2A_Q_LAST_ROLLOUT_KERNEL", _ffi.Q_LAST_ROLLOUT_KERNEL), ("W2A_Q_LAST_STEP_KERNEL", _ffi.Q_LAST_STEP_KERNEL), ("W2A_Q_LOCKSTEP", _ffi.Q_LOCKSTEP),
                      ("W2A_ST_STALE_GRAPH", _ffi.ST_STALE_GRAPH), ("W2A_ABI_VERSION", _ffi.ABI_VERSION)):
        assert enum(name) == val, name
    assert enum("W2A_ABI_VERSION") == 18


def test_ffi_struct_layout_matches_header():
    # 6 pointers + 6 int32 + 32 int32 + 1 int32 (+4 pad) + 2 pointers + 1 int32 (+4 pad) + 1 pointer + 1 int32 (+4 pad)
    assert C.sizeof(_ffi.Tables) == 6 * 8 + (6 + 32 + 1) * 4 + 4 + 2 * 8 + 8 + 8 + 8
    assert C.sizeof(_ffi.StateView) == 16 * 8


def test_env_refuses_to_run_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from weather2alert_amd import HeatAlertVecEnv, synth, tables

    ct = tables.compile_from_synth(synth.make_synth("linear", n_fips=8, years=[2006], n_samples=2, seed=0))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HeatAlertVecEnv(4, tables=ct, device="cuda:0")
    with pytest.raises(RuntimeError):
        HeatAlertVecEnv(4, tables=ct, device="cpu")


def test_header_is_plain_c_and_links(lib, tmp_path):
    """include/w2a.h compiles as strict C99 and a C program links libw2a.so and calls the host-only entry
    points (this is the binding any non-Python host language would use)."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = tmp_path / "c_abi_check"
    libdir = os.path.dirname(_ffi.lib_path())
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_abi_check.c"), "-L", libdir, "-lw2a", f"-Wl,-rpath,{libdir}",
                    "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert f"sizeof(w2a_tables)={C.sizeof(_ffi.Tables)}" in r.stdout


def test_probe_programs_build(lib, tmp_path):
    """The two stand-alone measurement programs behind DESIGN.md §5 (tools/fabric_probe.hip: the step's traffic
    pattern without env logic; tools/abi_probe.cpp: the step kernels through the C ABI from a C++ host) still
    cross-compile for gfx950 -- the second one against the current header and library."""
    import subprocess

    hipcc = build.hipcc_path()
    libdir = os.path.dirname(_ffi.lib_path())
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-DPROBE_F64", "-DPROBE_DEP", "-DPROBE_DAYS",
                    "-DPROBE_RANDOM_DATA", os.path.join(ROOT, "tools", "fabric_probe.hip"), "-o",
                    str(tmp_path / "fabric_probe")], check=True, capture_output=True)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tools", "abi_probe.cpp"), "-L", libdir, "-lw2a", f"-Wl,-rpath,{libdir}", "-o",
                    str(tmp_path / "abi_probe")], check=True, capture_output=True)
