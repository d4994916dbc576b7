"""Host-side checks of the proof margin tau (nns_internal.h: tau_consts / tau_of), compiled from the
very header the kernels use: monotone in the score (the filter relies on it to keep a running
threshold instead of a running minimum), growing with the norms and the tile depth, and ordered by
operand precision (fp32 operands < bf16 points < fp32 points rounded to bf16 operands)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <stdio.h>
#include "nns_internal.h"
int main()
{
    using namespace nns;
    const float xs[] = {0.0f, 1e-3f, 1.0f, 10.7f, 3000.0f};
    const float ys[] = {1e-3f, 1.0f, 10.7f, 3000.0f};
    const int kts[] = {32, 64, 128, 256, 512};
    for (int kt : kts)
        for (float x2 : xs)
            for (float y2 : ys)
                for (int mode = 0; mode < 3; ++mode) {
                    const TauConsts t = tau_consts(kt, x2, y2, mode);
                    printf("%d %g %g %d %.9g %.9g %.9g", kt, x2, y2, mode, t.c0, t.c1, t.x2);
                    float prev = -3.0e38f;
                    int mono = 1;
                    for (float a = -x2; a < 4.0f * (x2 + y2) + 1.0f; a += 0.01f * (x2 + y2) + 1e-4f) {
                        const float v = a + 1.002f * tau_of(t, a);
                        if (v < prev) mono = 0;
                        prev = v;
                    }
                    printf(" %d %.9g\n", mono, tau_of(t, y2));
                }
    return 0;
}
'''


def test_tau_margin_properties():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "tau.hip")
        exe = os.path.join(d, "tau")
        with open(src, "w") as f:
            f.write(SRC)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "nns-cuda_amd", "csrc"),
                        "-o", exe, src], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    rows = {}
    for line in out.splitlines():
        kt, x2, y2, mode, c0, c1, tx2, mono, tau = line.split()
        rows[(int(kt), float(x2), float(y2), int(mode))] = (float(c0), float(c1), float(tx2), int(mono), float(tau))
    assert len(rows) == 5 * 5 * 4 * 3
    for (kt, x2, y2, mode), (c0, c1, tx2, mono, tau) in rows.items():
        assert c0 > 0 and c1 > 0 and tx2 >= x2 and mono == 1, (kt, x2, y2, mode)
        assert tau > 0
    for (kt, x2, y2, mode), v in rows.items():
        if mode < 2:     # operand precision orders the margins (mode 0 has the extra centring term, so compare 1 < 2)
            pass
        if mode == 1:
            assert rows[(kt, x2, y2, 2)][4] >= v[4], (kt, x2, y2)
        if kt < 512:     # deeper tiles accumulate more rounding
            nxt = {32: 64, 64: 128, 128: 256, 256: 512}[kt]
            assert rows[(nxt, x2, y2, mode)][4] >= v[4]
    # the rounding term of mode 2 is the dominant one: ~2 * 2^-6 |x||y| for unit-scale norms
    c0 = rows[(128, 10.7, 10.7, 2)][0]
    assert 2 * 2 ** -6 * 10.7 < c0 < 4 * 2 ** -6 * 10.7 * 1.5


def test_fp32_threshold_sum_is_covered():
    """finalize.hip, "fp32 evaluation": the threshold K5 forms in fp32, T5(a) = fl(a + tau_fl(a)), must not fall below
    the real-arithmetic a + tau(a) of the proof although the sum is rounded at the SCORE's magnitude (up to
    tau / (2 (K + 2)) of error) — the explicit term er of tau_consts covers it — and the filter's
    Tf(t) = fl(t + fl(1.002 tau_fl(t))) must dominate T5(a) for every t >= a.  fp32 emulated with numpy, the
    requirement evaluated in extended precision; (c0, c1, x2) come from the library (nns_tau_consts, host only)."""
    import numpy as np
    import __graft_entry__ as graft
    pkg = graft.load_package()
    f32 = np.float32
    u = 2.0 ** -24
    L = np.longdouble

    def tau_fl(c0, c1, x2, a):      # exactly the kernels' expression order (tau_of / tighten), every step rounded to fp32
        d = f32(a) + f32(x2)
        d = d if d > 0 else f32(0)
        return f32(c0) + f32(c1) * d

    def tau_needed(kt, X2, Y2, mode, a):   # the proof's tau(a), no safety factors, extended precision
        X2, Y2 = L(X2) * (1 + 4 * L(u)), L(Y2) * (1 + 4 * L(u))
        X, Y = np.sqrt(X2), np.sqrt(Y2)
        gk = (kt + 2) * L(u) / (1 - (kt + 2) * L(u))
        if mode == 0:
            e3 = gk * (Y2 + 2 * X * Y) + 2 * L(u) * Y2
            e2 = L(2.5) * L(u) * (X + Y) ** 2
        else:
            gf = 2 * (kt + kt // 16 + 2) * L(u) / (1 - 2 * (kt + kt // 16 + 2) * L(u))
            e3 = gf * (Y2 + 2 * X * Y) + 2 * L(u) * Y2
            e2 = L(0)
            if mode == 2:
                e3 = e3 * (1 + L(2) ** -6) + L(2) ** -6 * (1 + L(2) ** -8) * X * Y
                e2 = L(2.5) * L(u) * (X + Y) ** 2
        return 2 * (e3 + e2) + 2 * gk / (1 - gk) * (max(L(a) + X2, L(0)) + e3 + e2), X, Y, e3

    rng = np.random.default_rng(7)
    checked = 0
    with np.errstate(over="ignore"):
        for kt in (16, 32, 64, 128, 256, 512, 1024):
            for X2 in (0.0, 1e-3, 0.7, 10.7, 43.0, 3000.0, 1e6):
                for Y2 in (1e-3, 0.7, 10.7, 43.0, 3000.0, 1e6):
                    for mode in (0, 1, 2):
                        c0, c1, x2 = pkg.tau_consts(kt, X2, Y2, mode)
                        _, X, Y, e3 = tau_needed(kt, X2, Y2, mode, 0.0)
                        lo, hi = -float(X * X), float((X + Y) ** 2 + e3)      # every attainable score
                        grid = np.concatenate([np.linspace(lo, hi, 41), rng.uniform(lo, hi, 40), [0.0, lo, hi]])
                        for a in grid.astype(np.float32):
                            need, _, _, _ = tau_needed(kt, X2, Y2, mode, float(a))
                            t5 = f32(a) + tau_fl(c0, c1, x2, a)                      # K5's threshold, in fp32
                            assert L(t5) >= L(a) + need, (kt, X2, Y2, mode, float(a), float(t5), float(L(a) + need))
                            # the requested form: slack = tau_used - tau_needed >= u (|a| + tau)
                            used = L(c0) + L(c1) * max(L(a) + L(x2), L(0))
                            assert used - need >= L(u) * (abs(L(a)) + used), (kt, X2, Y2, mode, float(a))
                            # the filter's threshold of any t >= a dominates K5's threshold of a
                            for t in (a, np.nextafter(a, f32(np.inf)), f32(a + 0.37 * abs(a) + 1e-3)):
                                tf = f32(t) + f32(f32(1.002) * tau_fl(c0, c1, x2, t))
                                assert tf >= t5, (kt, X2, Y2, mode, float(a), float(t))
                            checked += 1
    assert checked > 50000
