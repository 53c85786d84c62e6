#!/usr/bin/env python3
"""Development aid: kernel time of the eight_schools adaptation warmup (1000 iterations), repeated
so that run-to-run clock noise of a one-workgroup kernel can be told from a real change. With a
library built with -DEXMC_PROFILE_SECTIONS=1 (EXMC_HIP_LIB=...) the per-section cycle split of the
tree wave is printed by the library itself.  python tools/prof_warmup.py [repeats] [model]"""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, "tests")
import bench  # noqa: E402
from exmc_amd import sampler  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
name = sys.argv[2] if len(sys.argv) > 2 else "eight_schools"
spec, _ = bench.make_spec(name)
comp = sampler.compile(spec)
sampler.warmup(comp, spec.default_init, dict(num_warmup=2, seed=42))
ms = []
for _ in range(reps):
    t = sampler.warmup(comp, spec.default_init, dict(num_warmup=1000, seed=42))
    ms.append(comp.last_kernel_ms)
print("   in order: " + " ".join("%.2f" % v for v in ms), file=sys.stderr)
ms.sort()
print("%s eps %.17g  kernel ms: min %.2f  median %.2f  max %.2f" % (
    name, t["epsilon"], ms[0], ms[len(ms) // 2], ms[-1]), file=sys.stderr)
