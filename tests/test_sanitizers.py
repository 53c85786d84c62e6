"""Sanitizers on the CPU build (the GPU pool offers none): tools/sanitize_cpu.sh in
its quick form -- ASan + UBSan over both NIF shims and the host-side shims driven by their own CPU
tests, then ThreadSanitizer and ASan over HipNative.stream_run's sender / reaper / compare-and-swap
logic against a stub libexmc_hip (tests/host/stub_exmc_hip.c, tests/host/tsan_stream_driver.c).
The full form (the checker's sampler tests under ASan too) is what profiles/r5_sanitize/ holds."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have(lib):
    p = subprocess.run(["gcc", "-print-file-name=" + lib], capture_output=True, text=True).stdout.strip()
    return os.path.isabs(p) and os.path.exists(p)


@pytest.mark.skipif(not (_have("libasan.so") and _have("libtsan.so")), reason="libasan / libtsan not installed")
def test_sanitize_cpu_quick(tmp_path):
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize_cpu.sh"), str(tmp_path), "quick"],
                       capture_output=True, text=True, timeout=900)
    logs = "\n".join("== %s\n%s" % (f, open(os.path.join(tmp_path, f)).read()[-3000:]) for f in sorted(os.listdir(tmp_path)))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:] + logs
    assert open(tmp_path / "summary.txt").read().strip() == "sanitize_cpu: clean"
    for f in ("stream_driver_thread.log", "stream_driver_address_undefined.log"):
        txt = open(tmp_path / f).read()
        assert "tsan_stream_driver: ok" in txt and "Sanitizer" not in txt, txt[-2000:]
    assert " passed" in open(tmp_path / "asan_ubsan_pytest.log").read()
