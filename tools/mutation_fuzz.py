#!/usr/bin/env python3
"""Does the GPU call-sequence fuzz (tools/sequence_fuzz.py) catch a broken validity flag END TO END? Builds libw2a.so
from mutants of csrc/w2a_bookkeeping.h -- the list tests/test_bookkeeping_cpu.py uses on the CPU, one rule broken each --
and runs the fuzz on each until its first failing sequence.

    python tools/mutation_fuzz.py build [--jobs 4]          (anywhere with hipcc; writes weather2alert_amd/_lib/mutants/)
    python tools/mutation_fuzz.py run [--sequences 600]     (on the GPU box)

A mutant that survives here is not necessarily a hole: the CPU harness models budgets above 65 535 and recorded graphs
far more often than real sequences meet them, and catches all 28; this run says which of them ALSO change what a user
of the GPU library would see within a few hundred random sequences."""
import argparse
import concurrent.futures as cf
import importlib.util
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "weather2alert_amd", "_lib", "mutants")


def mutants():
    spec = importlib.util.spec_from_file_location("tb", os.path.join(ROOT, "tests", "test_bookkeeping_cpu.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.MUTANTS


def build_one(k, name, old, new):
    from weather2alert_amd import build as b

    csrc = os.path.join(ROOT, "weather2alert_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        dst = os.path.join(td, "csrc")
        shutil.copytree(csrc, dst)
        hp = os.path.join(dst, "w2a_bookkeeping.h")
        txt = open(hp).read()
        assert txt.count(old) == 1, (k, name)
        open(hp, "w").write(txt.replace(old, new))
        out = os.path.join(OUT, f"libw2a_m{k:02d}.so")
        r = subprocess.run([b.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-I{b.INC}",
                            os.path.join(dst, "w2a_kernels.hip"), "-o", out], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"mutant {k}: hipcc failed\n{r.stderr[-2000:]}")
    return k, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["build", "run"])
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--sequences", type=int, default=600)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--only", type=int, nargs="*", default=None)
    a = ap.parse_args()
    ms = mutants()
    ids = list(range(len(ms))) if a.only is None else a.only
    if a.what == "build":
        os.makedirs(OUT, exist_ok=True)
        with cf.ThreadPoolExecutor(a.jobs) as ex:
            for k, out in ex.map(lambda k: build_one(k, *ms[k]), ids):
                print(f"built mutant {k:2d}: {ms[k][0]}", flush=True)
        return 0
    caught = survived = 0
    for k in ids:
        lib = os.path.join(OUT, f"libw2a_m{k:02d}.so")
        if not os.path.exists(lib):
            print(f"mutant {k:2d}: library missing ({lib})")
            continue
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sequence_fuzz.py"), "--sequences", str(a.sequences),
                            "--seed", str(a.seed)], env=dict(os.environ, W2A_LIB=lib), capture_output=True, text=True)
        if r.returncode != 0:
            first = [ln for ln in r.stdout.splitlines() if ln.startswith("FAILED sequence")]
            why = [ln for ln in r.stdout.splitlines() if ln.startswith(">>>")]
            print(f"mutant {k:2d} CAUGHT   in {time.time() - t0:4.0f} s  {ms[k][0]}\n      {first[0].split('(')[0].strip() if first else '?'}: "
                  f"{(why[0][4:170] if why else r.stderr[-200:])}", flush=True)
            caught += 1
        else:
            print(f"mutant {k:2d} survived {a.sequences} sequences ({time.time() - t0:.0f} s)  {ms[k][0]}", flush=True)
            survived += 1
    print(f"mutation_fuzz: {caught} of {caught + survived} mutants caught by the GPU fuzz within {a.sequences} sequences (seed {a.seed})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
