#!/usr/bin/env python3
"""Instruction census of the filter kernels' hot blocks: per basic block with >= 16 MFMAs, how many vector (non-MFMA),
scalar, LDS and global instructions ride along.  (SQ_INSTS_VALU counts MFMAs as vector instructions: a kernel's
"VALU per MFMA" from the counters is 1 + this ratio.)   usage: isa_census.py [kernel-name-substring ...]"""
import collections, importlib.util, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("c", os.path.join(ROOT, "tools", "check_mfma_hazards.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
lines = open(m.compile_isa()).read().splitlines()
want = sys.argv[1:] or ["OpBF16TILi16ELi8", "OpF32TILi16ELi2"]
for sym in [l.split(":")[0] for l in lines if l.startswith("_ZN3nns13filter_kernel") and "@" in l]:
    if not any(w in sym for w in want):
        continue
    start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur, name = [], [], "entry"
    for l in lines[start:end]:
        mm = re.match(r"^(\.LBB\S+):", l)
        if mm:
            blocks.append((name, cur)); cur = []; name = mm.group(1)
        else:
            cur.append(l)
    blocks.append((name, cur))
    print(sym[20:70])
    tot_m = tot_v = 0
    for name, b in blocks:
        ops = collections.Counter(t[0] for t in (x.strip().split() for x in b) if t and not t[0].startswith((";", ".")))
        nm = sum(c for o, c in ops.items() if o.startswith("v_mfma"))
        if nm < 16:
            continue
        valu = sum(c for o, c in ops.items() if o.startswith("v_") and not o.startswith("v_mfma"))
        tot_m += nm; tot_v += valu
        top = ", ".join(f"{c} {o}" for c, o in sorted(((c, o) for o, c in ops.items() if o.startswith("v_") and not o.startswith("v_mfma")), reverse=True)[:6])
        print(f"  {name:12s} mfma {nm:4d}  other vector {valu:4d}  scalar {sum(c for o, c in ops.items() if o.startswith('s_')):4d}"
              f"  ds {sum(c for o, c in ops.items() if o.startswith('ds_')):3d}  global {sum(c for o, c in ops.items() if o.startswith('global_')):3d}   [{top}]")
    if tot_m:
        print(f"  hot blocks in all: {tot_v} other vector instructions per {tot_m} MFMAs = {tot_v / tot_m:.2f}")
