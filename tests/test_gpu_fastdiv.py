"""The short f64 division used by the kernels (include/exmc_detmath.h: exmc_rcp_refined +
exmc_div_core) against the compiler's IEEE expansion of `/` on the device: 5 x 2^28 quotients over
the operand ranges the kernels claim for it, bit for bit (tools/probe/fastdiv_probe.hip)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_short_division_is_ieee_division(hip):
    src = os.path.join(ROOT, "tools", "probe", "fastdiv_probe.hip")
    exe = os.path.join(ROOT, "tools", "probe", "fastdiv_probe")
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off",
                        "-I", os.path.join(ROOT, "include"), "-o", exe, src], check=True)
    out = subprocess.run([exe], check=False, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if "quotients" in ln]
    assert len(lines) == 5 and all(ln.endswith(" 0 mismatches") for ln in lines), out.stdout
