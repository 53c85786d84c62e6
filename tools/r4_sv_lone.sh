#!/bin/bash
# lone-wave rate of the sv sampling kernel (1024 chains = one wave per SIMD) with the cross-row
# stages of its 64-lane sums through the LDS crossbar (svx1) and through v_readlane (svx0)
out=gpurun_out/${1:-r4_sv_lone}; mkdir -p $out
for v in 1 0 1 0; do
  EXMC_HIP_LIB=$PWD/exmc_amd/lib/libexmc_hip_svx$v.so python3 bench.py --model sv --chains-per-gpu 1024 --steps 4 --no-cpu --no-multi-step > $out/x$v.json 2> $out/x$v.err || { tail -3 $out/x$v.err; exit 1; }
  python3 -c "import json; d=json.load(open('$out/x$v.json')); print('xrow_lds=$v lone: %.4e lf/s kernel %.1f ms lf %d' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['leapfrogs_per_launch']))"
done
for v in 1 0; do
  EXMC_HIP_LIB=$PWD/exmc_amd/lib/libexmc_hip_svx$v.so python3 bench.py --model sv --no-cpu --no-multi-step > $out/full$v.json 2> $out/full$v.err || { tail -3 $out/full$v.err; exit 1; }
  python3 -c "import json; d=json.load(open('$out/full$v.json')); print('xrow_lds=$v 2048x1000: %.4e lf/s kernel %.1f ms lf %d' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['leapfrogs_per_launch']))"
done
