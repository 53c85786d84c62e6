#!/bin/bash
# round 6: logistic-only development builds (-DEXMC_DEV_ONLY=3) alternating on one box, bench lines only
# (a library built from another numeric contract than the checker's cannot run the parity tests).
#   gpurun -- 'bash tools/r6_lg_ab.sh <tag> lib1.so lib2.so ...'
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2: %.4e lf/s kernel %.2f ms adapt %.4f s eps %.17g lf %d  ns/lf %.4f' % (d['value'], d['roofline']['kernel_ms'], d['ess_wall_s']['adaptation'], d['step_size'], d['roofline']['leapfrogs_per_launch'], 1e6 * d['roofline']['kernel_ms'] / d['roofline']['leapfrogs_per_launch']))"; }
for i in 1 2 3; do
  for lib in "$@"; do
    n=$(basename $lib .so)
    EXMC_HIP_LIB=$PWD/$lib python3 bench.py --model logistic --no-cpu --no-multi-step > $out/$n.$i.json 2> $out/$n.$i.err || { tail -5 $out/$n.$i.err; exit 1; }
    line $out/$n.$i.json $n
  done
done
