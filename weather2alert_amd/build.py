"""Build the in-tree HIP library (libw2a.so) for gfx950 with hipcc.

``python -m weather2alert_amd.build`` or ``build_lib()``. hipcc cross-compiles without a
GPU; the resulting .so is git-ignored but travels with the working tree to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = os.path.join(PKG, "csrc", "w2a_kernels.hip")
INC = os.path.join(ROOT, "include")
LIB_DIR = os.path.join(PKG, "_lib")
LIB = os.path.join(LIB_DIR, "libw2a.so")


def hipcc_path() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need the ROCm toolchain to build libw2a.so)")


HASH_FILE = LIB + ".srchash"  # sidecar written by build_lib(): what the library was built from


def _dep_files() -> list:
    csrc = os.path.dirname(SRC)
    return sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hip.h", ".h"))) + \
        [os.path.join(INC, "w2a.h")]


def build_hash() -> str:
    """sha256 over every source the library is compiled from (names + contents) and the extra compiler flags."""
    import hashlib

    h = hashlib.sha256()
    for path in _dep_files():
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    h.update(os.environ.get("W2A_CXXFLAGS", "").encode())
    return h.hexdigest()


def needs_build() -> bool:
    """The library is missing, or was built from other sources than the ones in the tree. Decided by content hash (the
    sidecar build_lib() writes), not by file times: a fresh checkout, a copy or a read-only install whose sources merely
    LOOK newer than a correct prebuilt library must load it as it is. A library without a sidecar (built by hand) falls
    back to the file-time comparison."""
    if not os.path.exists(LIB):
        return True
    if os.path.exists(HASH_FILE):
        try:
            return open(HASH_FILE).read().strip() != build_hash()
        except OSError:
            return True
    m = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > m for d in _dep_files())


STEP_SOURCES = ("w2a_common.hip.h", "w2a_step.hip.h", "w2a_step64.hip.h", "w2a_step_dispatch.hip.h", "w2a_bookkeeping.h")


def source_sha(files: tuple = STEP_SOURCES) -> str:
    """sha256 over the sources the step kernels and their dispatch are built from (file names + contents): identifies
    the build a step-kernel profile was collected on, independently of commits that do not touch them."""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.dirname(SRC)
    for path in [os.path.join(csrc, f) for f in sorted(files)]:
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    tmp = f"{LIB}.{os.getpid()}.tmp"  # several ranks may build at once: private temp + atomic rename
    # what is about to be compiled, hashed BEFORE the compiler reads it: an edit during the compile then leaves a sidecar
    # that does not match the tree, and the next needs_build() rebuilds
    src_hash = build_hash()
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-I{INC}", SRC,
           "-o", tmp]
    cmd += os.environ.get("W2A_CXXFLAGS", "").split()  # e.g. -DLANES=8 for kernel-geometry A/B runs
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed ({r.returncode}):\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(r.stderr)
    # sidecar away first, library in, sidecar back: at no moment does an OLD sidecar stand beside the NEW library (or
    # the other way round) claiming it current -- a crash in between leaves a library without sidecar = file-time rule
    try:
        os.remove(HASH_FILE)
    except OSError:
        pass
    os.replace(tmp, LIB)
    htmp = f"{HASH_FILE}.{os.getpid()}.tmp"
    with open(htmp, "w") as f:
        f.write(src_hash + "\n")
    os.replace(htmp, HASH_FILE)
    return LIB


PROBE_SRC = os.path.join(ROOT, "tools", "fabric_probe.hip")
PROBE_LIB = os.path.join(LIB_DIR, "libw2a_probe.so")
# the flag set of profiles/probe_ceiling.json: the step's fp64 chain, gather indices through the streamed state, another
# day slice every launch, random data, state updated in place, the 16-B packed lock-step streams
PROBE_FLAGS = ("-DPROBE_LIB", "-DPROBE_F64", "-DPROBE_DEP", "-DPROBE_DAYS", "-DPROBE_RANDOM_DATA", "-DPROBE_INPLACE")


def build_probe_lib(force: bool = False, packed: bool = True) -> str:
    """MEASUREMENT ONLY (not part of the env): tools/fabric_probe.hip as a shared library, so that bench.py can time the
    arithmetic-free access pattern of the step and a plain float4 copy inside its own process, on its own box."""
    out = PROBE_LIB if packed else PROBE_LIB.replace(".so", "_unpacked.so")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(PROBE_SRC):
        return out
    os.makedirs(LIB_DIR, exist_ok=True)
    tmp = f"{out}.{os.getpid()}.tmp"
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *PROBE_FLAGS,
           *(["-DPROBE_PACKED"] if packed else []), PROBE_SRC, "-o", tmp]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on the probe library ({r.returncode}):\n{r.stdout}\n{r.stderr}")
    os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
    print(build_probe_lib(force="--force" in sys.argv))
    print(build_probe_lib(force="--force" in sys.argv, packed=False))
