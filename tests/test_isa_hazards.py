"""The bf16 filter issues its MFMAs from inline asm, where hipcc adds no hazard wait states: the
generated ISA is re-checked after every build (tools/check_mfma_hazards.py)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _checker():
    spec = importlib.util.spec_from_file_location("check_mfma_hazards", os.path.join(ROOT, "tools", "check_mfma_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_checker_flags_an_early_read_and_accepts_in_place_accumulation():
    chk = _checker()
    early = """
    v_mfma_f32_16x16x32_bf16 v[178:181], v[232:235], v[62:65], v[178:181]
    v_mfma_f32_16x16x32_bf16 v[170:173], v[232:235], v[94:97], v[170:173]
    v_min3_f32 v163, v185, v178, v179
    """
    lines = list(enumerate(early.splitlines(), 1))
    assert len(chk.check_kernel("k", lines)) == 1
    ok = """
    v_mfma_f32_16x16x32_bf16 v[178:181], v[232:235], v[62:65], v[178:181]
    v_mfma_f32_16x16x32_bf16 v[178:181], v[236:239], v[66:69], v[178:181]
    s_nop 7
    v_min3_f32 v163, v185, v178, v179
    """
    assert chk.check_kernel("k", list(enumerate(ok.splitlines(), 1))) == []
    overlap = """
    v_mfma_f32_16x16x32_bf16 v[178:181], v[232:235], v[62:65], v[178:181]
    v_mfma_f32_16x16x32_bf16 v[190:193], v[232:235], v[62:65], v[180:183]
    """
    assert len(chk.check_kernel("k", list(enumerate(overlap.splitlines(), 1)))) == 1


def test_filter_kernels_have_no_mfma_read_hazards():
    chk = _checker()
    path = chk.compile_isa()
    with open(path) as f:
        text = f.read().splitlines()
    import re
    kernels, cur = {}, None
    for i, l in enumerate(text, 1):
        m = re.match(r"^(_Z\w*filter_kernel\w*):", l)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is not None:
            if "s_endpgm" in l:
                cur = None
                continue
            cur.append((i, l))
    assert len(kernels) == 12, list(kernels)   # fp32 KT = 16 / 32 / 64 / 128 / 256, bf16 KT = 128 / 256 / 384 / 512 / 640 / 768 / 1024
    for name, lines in kernels.items():
        assert sum("v_mfma" in l for _, l in lines) >= 64, name   # (every interval is fully unrolled: 32 steps)
        assert chk.check_kernel(name, lines) == [], name
        # no register spills in the hot kernels: a scratch reload sits behind s_waitcnt vmcnt(0),
        # which would also drain the LDS-DMA ring
        assert not any("scratch_" in l for _, l in lines), name
