#!/usr/bin/env python3
"""Register / spill / v_mov report of the filter kernels' ISA (tuning aid)."""
import importlib.util, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("c", os.path.join(ROOT, "tools", "check_mfma_hazards.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
path = m.compile_isa()
txt = open(path).read()
for b in re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, flags=re.S):
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    if "filter_kernel" not in name:
        continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
    print(name[20:62], "vgpr", g("vgpr_count"), "sgpr", g("sgpr_count"), "spill", g("vgpr_spill_count"), "scratch", g("private_segment_fixed_size"))
lines = txt.splitlines()
for sym in [l.split(":")[0] for l in lines if l.startswith("_ZN3nns13filter_kernel") and "@" in l]:
    start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    run = best = 0
    for l in body:
        if "v_mov_b32" in l:
            run += 1; best = max(best, run)
        else:
            run = 0
    print(sym[20:62], len(body), "lines, v_mov", sum("v_mov_b32" in l for l in body), "longest run", best,
          "mfma", sum("v_mfma" in l for l in body), "barriers", sum("s_barrier" in l for l in body))
print(path)
