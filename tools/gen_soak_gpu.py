#!/usr/bin/env python3
"""GPU soak of the lane layout: the random hierarchical models of tests/test_codegen_lanes.py for
more seeds than the test suite runs (bit-exact log-density, gradient and a short sample/3 against
the generated text on the CPU).   gpurun -- 'python tools/gen_soak_gpu.py 3 15'
With --plate: the random models of tests/test_gpu_codegen_random.py instead (the 16-lane plate layout and one lane per
chain).   gpurun -- 'python tools/gen_soak_gpu.py 10 40 --plate'"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_codegen_lanes as T  # noqa: E402
import test_gpu_codegen_random as TR  # noqa: E402
from exmc_amd import _lib  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
hip = _lib.load()
bad = []
for seed in range(lo, hi):
    try:
        if "--plate" in sys.argv:
            TR.test_random_generated_model_bit_exact(hip, seed)
        else:
            T.test_random_models_in_the_lane_layout_bit_exact(seed, hip)
        print(seed, "ok", flush=True)
    except Exception as e:   # noqa: BLE001
        print(seed, "FAIL", repr(e)[:300], flush=True)
        bad.append(seed)
print("failed seeds:", bad)
sys.exit(1 if bad else 0)
