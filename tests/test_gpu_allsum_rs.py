"""The reduce-scatter forms of the 64-lane sums (exmc_device.hpp rs64_*) against the butterfly
group_allsum_n<64, N> on the device, bit for bit (tools/probe/allsum_rs_probe.hip): the sum tree of
every total is the butterfly's, only the lane that performs an addition differs."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reduce_scatter_sums_equal_the_butterfly(hip):
    src = os.path.join(ROOT, "tools", "probe", "allsum_rs_probe.hip")
    exe = os.path.join(ROOT, "tools", "probe", "allsum_rs_probe")
    dep = os.path.join(ROOT, "exmc_amd", "csrc", "exmc_device.hpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(dep)):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-I",
                        os.path.join(ROOT, "include"), "-o", exe, src], check=True, capture_output=True)
    out = subprocess.run([exe], check=False, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout + out.stderr
