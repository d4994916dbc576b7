#!/usr/bin/env python3
"""Static check of the filter kernels' ISA for MFMA read-after-write hazards.

The bf16 filter issues v_mfma_f32_16x16x32_bf16 from inline asm (so that accumulators stay in
place, filter_mfma.hip OpBF16).  hipcc does not know an asm statement is an MFMA: it inserts no
wait states behind it and is free to schedule readers of the result right after it, and the
hardware does not interlock — a too-early read returns stale data.  The kernel keeps the
readers a whole step behind the writers by construction; this script re-checks that on the
generated code after every build:

  for every MFMA, none of the following WAIT instructions-worth of wait states may contain a
  non-MFMA instruction that reads or writes the MFMA's destination registers, nor an MFMA
  that reads them as A/B operand or as a DIFFERENT (partially overlapping) C tuple.
  (An MFMA accumulating in place on exactly the same tuple is the supported back-to-back form.)

WAIT = 8 for the 4-pass 16x16x32 (what hipcc itself inserts behind the builtin: s_nop 7), 12 for
the 32x32x16 / 32x32x2 forms (s_nop 11).  s_nop N counts N + 1 wait states, every other
instruction 1 (a lower bound on the time it takes).  Straight-line scan per kernel; labels do
not reset it (fall-through is the worst case), an unconditional branch does, conditional branches
are not followed (their targets start with no MFMA pending: optimistic, so the kernel source keeps
its readers a full step behind the writers on every path).

usage: check_mfma_hazards.py [file.s]      (without a file: compiles filter_mfma.hip to ISA)
exit status 0 = clean.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nns-cuda_amd", "csrc")

WAITS = {"v_mfma_f32_16x16x32_bf16": 8, "v_mfma_f32_32x32x16_bf16": 12, "v_mfma_f32_32x32x2_f32": 12,
         "v_mfma_f32_16x16x4_f32": 8}


def compile_isa() -> str:
    out = os.path.join(tempfile.mkdtemp(prefix="nns_isa_"), "filter_mfma.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-fno-honor-nans", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out,
           os.path.join(CSRC, "filter_mfma.hip")]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def vregs(tok: str):
    """VGPR numbers named by one operand token ('v12', 'v[4:7]'); empty for anything else."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    return set()


def operands(line: str):
    body = line.split(";")[0].strip()
    parts = body.split(None, 1)
    if len(parts) < 2:
        return parts[0] if parts else "", []
    ops = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", parts[1])]
    return parts[0], ops


def check_kernel(name: str, lines) -> list:
    problems = []
    pending = []   # [dst set, remaining wait states, mnemonic, line number, exact dst token]
    for ln, raw in lines:
        s = raw.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        mn, ops = operands(s)
        if not mn or mn.startswith("."):
            continue
        if mn in ("s_branch", "s_setpc_b64", "s_endpgm"):   # control leaves: what follows is not reached from here
            pending = []
            continue
        if mn == "s_nop":
            n = int(ops[0], 0) + 1 if ops else 1
            for p in pending:
                p[1] -= n
            pending = [p for p in pending if p[1] > 0]
            continue
        touched = set()
        for o in ops:
            touched |= vregs(o)
        is_mfma = mn.startswith("v_mfma")
        for p in pending:
            if not (touched & p[0]):
                continue
            if is_mfma:
                # allowed: accumulate in place on exactly the same tuple (dst == srcC == pending dst)
                ab = vregs(ops[1]) | vregs(ops[2])
                same_c = ops[3] == p[4] and ops[0] == p[4]
                if (ab & p[0]) or not same_c:
                    problems.append(f"{name}: line {ln}: '{s}' uses {p[4]} of {p[2]} (line {p[3]}) "
                                    f"{WAITS.get(p[2], 12) - p[1]} wait states after it")
            else:
                problems.append(f"{name}: line {ln}: '{s}' touches {p[4]} written by {p[2]} (line {p[3]}) "
                                f"only {WAITS.get(p[2], 12) - p[1]} wait states earlier")
        for p in pending:
            p[1] -= 1
        pending = [p for p in pending if p[1] > 0]
        if is_mfma:
            pending.append([vregs(ops[0]), WAITS.get(mn, 12), mn, ln, ops[0]])
    return problems


def main() -> int:
    path = sys.argv[1] if len(sys.argv) > 1 else compile_isa()
    with open(path) as f:
        text = f.read().splitlines()
    kernels, cur, name = {}, None, None
    for i, l in enumerate(text, 1):
        m = re.match(r"^(_Z\w*filter_kernel\w*):", l)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
            continue
        if cur is not None:
            if "s_endpgm" in l:
                cur = None
                continue
            cur.append((i, l))
    if not kernels:
        print("no filter kernels found in", path)
        return 2
    bad = []
    for name, lines in kernels.items():
        n_mfma = sum("v_mfma" in l for _, l in lines)
        p = check_kernel(name, lines)
        print(f"{name}: {n_mfma} MFMAs, {len(p)} hazard(s)")
        bad += p
    for b in bad[:40]:
        print("  " + b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
