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
