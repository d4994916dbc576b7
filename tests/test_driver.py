"""GPU test of the main.cu-style driver (nns-cuda_amd/nns_driver): the C++ shim
mi355x::cudaCall through the reference harness' own samples and data recipe."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_driver_reference_samples(built, orc, golden_dir):
    exe = os.path.join(ROOT, "nns-cuda_amd", "nns_driver")
    assert os.path.exists(exe)
    z = np.load(f"{golden_dir}/golden_recipe.npz")
    # are we on the glibc the fixtures were drawn with?
    k, m, n, q, r = next(orc.ref_recipe([(3, 1, 1024)], seed=1000))
    if orc.fnv1a64(q) != int(z["s0_input_fnv"][0]):
        pytest.skip("libc rand() stream differs from the fixture's")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)     # all ten samples of main.cu:38-51
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("CudaCall")]
    assert len(lines) == 10
    for i, ln in enumerate(lines):
        mm = re.match(r"CudaCall 100,\s*(\d+),\s*(\d+),\s*(\d+),\s*([\d.]+)ms\s+first=(\d+) fnv=([0-9a-f]+)", ln)
        assert mm, ln
        assert (int(mm[1]), int(mm[2]), int(mm[3])) == tuple(int(v) for v in z[f"s{i}_shape"])
        want = z[f"s{i}_idx"]
        assert int(mm[5]) == int(want[0])
        assert int(mm[6], 16) == orc.fnv1a64(want), f"sample {i}"


def test_driver_c1_shape_through_the_shim(built, orc):
    """BASELINE's C1 (1024 queries x 4096 refs x 3-D, the reference's own CPU-runnable case) through
    mi355x::cudaCall — the function-pointer type of main.cu:7 — on the driver's data recipe, against V0."""
    exe = os.path.join(ROOT, "nns-cuda_amd", "nns_driver")
    k, m, n, q, r = next(orc.ref_recipe([(3, 1024, 4096)], seed=1000))
    want, _ = orc.v0_search(q, r)
    out = subprocess.run([exe, "--shape", "3,1024,4096"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("CudaCall")]
    assert len(lines) == 1
    mm = re.match(r"CudaCall 100,\s*3,\s*1024,\s*4096,\s*([\d.]+)ms\s+first=(\d+) fnv=([0-9a-f]+)", lines[0])
    assert mm, lines[0]
    assert int(mm[2]) == int(want[0]) and int(mm[3], 16) == orc.fnv1a64(want)
