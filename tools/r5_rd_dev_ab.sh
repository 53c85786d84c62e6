#!/bin/bash
# radon-only development builds (exmc_amd/lib/libexmc_hip_rd*.so, -DEXMC_DEV_ONLY=2) alternating on one box:
# bench lines (kernel ms, adaptation s, step size and leapfrog count -- equal bits show there).
#   gpurun -- 'bash tools/r5_rd_dev_ab.sh <tag> lib1.so lib2.so ...'
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2: %.4e lf/s kernel %.2f ms adapt %.4f s ess/s %.4e eps %.17g lf %d' % (d['value'], d['roofline']['kernel_ms'], d['ess_wall_s']['adaptation'], d['ess_per_s'], d['step_size'], d['roofline']['leapfrogs_per_launch']))"; }
for i in 1 2 3; do
  for lib in "$@"; do
    n=$(basename $lib .so)
    EXMC_HIP_LIB=$PWD/$lib python3 bench.py --model radon --no-cpu --no-multi-step > $out/$n.$i.json 2> $out/$n.$i.err || { tail -5 $out/$n.$i.err; exit 1; }
    line $out/$n.$i.json $n
  done
done
