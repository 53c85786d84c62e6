#!/usr/bin/env python3
"""Residency and per-wave rates from a wave-probe file (tools/sv_probe.sh):
    python tools/sv_probe_report.py gpurun_out/<tag>/waves_*.txt"""
import sys

import numpy as np


def report(path):
    f = open(path)
    f.readline()
    a = np.loadtxt(f)
    a = a[a[:, 3] > 0]
    wg, place, clk, lf, w0, w1 = a.T
    t0 = w0.min()
    s, e = (w0 - t0) / 1e5, (w1 - t0) / 1e5          # ms (100 MHz ticks)
    ts = np.linspace(0, e.max(), 12)
    res = [int(((s <= t) & (e > t)).sum()) for t in ts]
    old, young = wg < 1024, wg >= 1024
    print("%s\n  launch %.0f ms; wave-residency %.2f per SIMD on average; resident waves at 12 times: %s"
          % (path, e.max(), (e - s).sum() / e.max() / 1024, res))
    print("  older waves: end mean %.0f ms, %.0f leapfrogs/ms; younger: end mean %.0f ms, %.0f leapfrogs/ms"
          % (e[old].mean(), (lf[old] / e[old]).mean(), e[young].mean(), (lf[young] / e[young]).mean()))
    print("  leapfrogs per wave mean %.0f sd %.0f max/mean %.2f; balanced at the whole-launch rate: %.0f ms"
          % (lf.mean(), lf.std(), lf.max() / lf.mean(), lf.sum() / (lf.sum() / ((e - s).sum() / 2))))


for p in sys.argv[1:]:
    report(p)
