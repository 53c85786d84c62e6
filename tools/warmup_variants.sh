#!/bin/bash
# development: adaptation time of the bench under several builds of the library (lib/libexmc_dev_*.so)
out=gpurun_out/${1:-variants}; mkdir -p $out
for lib in exmc_amd/lib/libexmc_dev_*.so; do
  for rep in 1 2; do
    EXMC_HIP_LIB=$PWD/$lib python bench.py --no-cpu --no-multi-step > $out/v.json 2> $out/v.err || { tail -3 $out/v.err; exit 1; }
    python -c "
import json,sys;d=json.load(open('$out/v.json'));print('$lib', 'adaptation %.3f ms  nuts %.3f ms' % (d['ess_wall_s']['adaptation']*1e3, d['roofline']['kernel_ms']))"
  done
done
