#!/usr/bin/env python3
"""Static instruction budget of one kernel of libw2a.so from its gfx950 ISA (hipcc -S; cross-compiles, no GPU needed):
instructions per basic block, classified as the SQ counters classify them (MFMA / transcendental / other VALU / LDS /
VMEM / SALU+SMEM), with the loop nesting taken from the backward branches. The per-launch totals of
k_posterior_mean_i8 follow from these counts and the workload's tile counts (waves x set-up + tile-heads x loop body) and
are compared in DESIGN.md §5 with the rocprofv3 --pmc figures of profiles/r05/pmc_configs2_pm_matrix_i8.json
(SQ_INSTS_VALU, SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES) -- VERDICT r4 item 6.

usage: python tools/isa_budget.py [--kernel k_posterior_mean_i8] [--asm /tmp/w2a_dev.s]"""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from weather2alert_amd import build  # noqa: E402


def classify(op: str) -> str:
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier") or op.startswith("s_setprio"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def blocks_of(asm: str, kernel: str):
    """[(label, {class: count}, [branch targets])] of the kernel's body, in program order."""
    lines = asm.splitlines()
    start = next(i for i, ln in enumerate(lines) if re.match(r"^_Z\d+" + re.escape(kernel) + r"\w*:", ln))
    out, cur = [], ("entry", {}, [])
    for ln in lines[start + 1:]:
        if ln.startswith("\t.size") or ln.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            out.append(cur)
            cur = (m.group(1), {}, [])
            continue
        m = re.match(r"^\t([a-z_0-9]+)\s*(.*)$", ln)
        if not m or m.group(1).startswith("."):
            continue
        op, args = m.group(1), m.group(2)
        c = classify(op)
        cur[1][c] = cur[1].get(c, 0) + 1
        if c == "branch":
            t = re.search(r"(\.LBB\d+_\d+)", args)
            if t:
                cur[2].append(t.group(1))
    out.append(cur)
    return out


def loops_of(blocks):
    """Backward branches -> loops [first block index, last block index], innermost last."""
    pos = {b[0]: i for i, b in enumerate(blocks)}
    loops = []
    for i, b in enumerate(blocks):
        for t in b[2]:
            if t in pos and pos[t] <= i:
                loops.append((pos[t], i))
    return sorted(set(loops))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_posterior_mean_i8")
    ap.add_argument("--asm", default=None)
    a = ap.parse_args()
    if a.asm is None:
        a.asm = "/tmp/_w2a_dev.s"
        subprocess.run([build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{build.INC}", "--cuda-device-only", "-S",
                        build.SRC, "-o", a.asm], check=True, capture_output=True)
    blocks = blocks_of(open(a.asm).read(), a.kernel)
    loops = loops_of(blocks)
    depth = [sum(1 for lo, hi in loops if lo <= i <= hi) for i in range(len(blocks))]
    classes = ("mfma", "trans", "valu", "lds", "vmem", "smem", "salu", "branch", "wait")
    print(f"{a.kernel}: {len(blocks)} basic blocks, loops (block ranges) {loops}")
    per_depth = {}
    for (lab, cnt, _), d in zip(blocks, depth):
        acc = per_depth.setdefault(d, {c: 0 for c in classes})
        for c in classes:
            acc[c] += cnt.get(c, 0)
    for d in sorted(per_depth):
        print(f"  loop depth {d}: " + ", ".join(f"{c} {per_depth[d][c]}" for c in classes if per_depth[d][c]))
    # innermost loop bodies one by one (the row-tile loop is unrolled inside the column-tile loop)
    for lo, hi in loops:
        acc = {c: 0 for c in classes}
        for i in range(lo, hi + 1):
            for c in classes:
                acc[c] += blocks[i][1].get(c, 0)
        print(f"  loop blocks {lo}..{hi} ({blocks[lo][0]}): " + ", ".join(f"{c} {acc[c]}" for c in classes if acc[c]))
    json.dump({"kernel": a.kernel, "per_depth": per_depth, "loops": loops}, open("/tmp/isa_budget.json", "w"))


if __name__ == "__main__":
    main()
