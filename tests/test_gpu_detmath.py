"""Device spellings of the numeric contract (include/exmc_detmath.h: exp / log, their asm-core
forms and the range-restricted variants with v_ldexp_f64 scaling) against the host build of the
same header, bit for bit over 13 x 2^20 arguments (tools/probe/detmath_probe.hip)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_exp_log_equal_the_host_contract(hip):
    src = os.path.join(ROOT, "tools", "probe", "detmath_probe.hip")
    exe = os.path.join(ROOT, "tools", "probe", "detmath_probe")
    hdr = os.path.join(ROOT, "include", "exmc_detmath.h")
    tab = os.path.join(ROOT, "include", "exmc_logtab.h")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(tab)):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off",
                        "-I", os.path.join(ROOT, "include"), "-o", exe, src], check=True)
    out = subprocess.run([exe], check=False, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if "mismatches" in ln]
    assert len(lines) == 13 and all(ln.endswith(" 0 mismatches") for ln in lines), out.stdout
