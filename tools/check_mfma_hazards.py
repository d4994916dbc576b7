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

WAIT = 8 for the 4-pass 16x16x32 (what hipcc itself inserts behind the builtin: s_nop 7) — the
only MFMA the kernels issue from inline asm.  The 32x32x16 / 32x32x2 forms go through the
compiler builtins: hipcc's own hazard recognizer places their wait states (e.g. exactly 11 in
front of a VALU overwrite of an 8-pass result) and they are not re-checked here.
s_nop N counts N + 1 wait states, every other
instruction 1 (a lower bound on the time it takes).  The fall-through path is scanned from the
top of each kernel; every branch taken while results are pending (conditional, or the loop's
back-edge) is followed into its target for as long as they stay pending.

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

WAITS = {"v_mfma_f32_16x16x32_bf16": 8}   # the inline-asm MFMA; builtin forms are the compiler's business


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
    """lines: [(line number, text)].  Scans the fall-through path from the top and, from every
    conditional branch taken while MFMA results are still pending, the target's first
    instructions with that pending state (pending entries expire within <= 12 wait states, so
    these side scans are short)."""
    insts = []      # (line number, mnemonic, operand tokens, raw)
    label_at = {}   # label -> index into insts of the first instruction behind it
    for ln, raw in lines:
        s = raw.strip()
        if not s or s.startswith((";", "//")):
            continue
        m = re.match(r"^([.\w$]+):", s)
        if m:
            label_at[m.group(1)] = len(insts)
            continue
        if s.startswith("."):
            continue
        mn, ops = operands(s)
        if mn:
            insts.append((ln, mn, ops, s))

    problems = set()

    def scan(start: int, pending: list, follow: bool):
        side = []
        i = start
        while i < len(insts):
            ln, mn, ops, s = insts[i]
            i += 1
            if mn in ("s_branch", "s_setpc_b64", "s_endpgm"):
                if mn == "s_branch" and pending and ops and ops[0] in label_at:
                    side.append((label_at[ops[0]], [list(p) for p in pending]))
                pending = []
                if not follow:
                    break
                continue
            if mn.startswith("s_cbranch") and pending and ops and ops[-1] in label_at:
                side.append((label_at[ops[-1]], [list(p) for p in pending]))
            if mn == "s_nop":
                n = int(ops[0], 0) + 1 if ops else 1
                for p in pending:
                    p[1] -= n
                pending = [p for p in pending if p[1] > 0]
                if not follow and not pending:
                    break
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
                        problems.add(f"{name}: line {ln}: '{s}' uses {p[4]} of {p[2]} (line {p[3]}) "
                                     f"{WAITS.get(p[2], 12) - p[1]} wait states after it")
                else:
                    problems.add(f"{name}: line {ln}: '{s}' touches {p[4]} written by {p[2]} (line {p[3]}) "
                                 f"only {WAITS.get(p[2], 12) - p[1]} wait states earlier")
            for p in pending:
                p[1] -= 1
            pending = [p for p in pending if p[1] > 0]
            if is_mfma and mn in WAITS:   # (builtin MFMA forms: hipcc's hazard recognizer owns their wait states)
                pending.append([vregs(ops[0]), WAITS[mn], mn, ln, ops[0]])
            if not follow and not pending:
                break
        return side

    todo = scan(0, [], True)
    seen = 0
    while todo and seen < 100000:
        start, pend = todo.pop()
        seen += 1
        todo += scan(start, pend, False)
    return sorted(problems)


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
