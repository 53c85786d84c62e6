#!/bin/bash
# development: bench one model under several builds of the library
#   gpurun -- 'bash tools/model_variants.sh <tag> <model> <steps> "<extra bench args>" lib1.so lib2.so ...'
out=gpurun_out/$1; model=$2; steps=$3; extra=$4; shift 4; mkdir -p $out
for lib in "$@"; do
  EXMC_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --model $model --no-cpu --no-multi-step --steps $steps --warmup 1 $extra > $out/v.json 2> $out/v.err || { tail -3 $out/v.err; exit 1; }
  python -c "
import json;d=json.load(open('$out/v.json'));print('$lib $extra', 'lf/s %.4g  nuts %.2f ms  adaptation %.2f ms  mean lf/draw %.1f' % (d['value'], d['roofline']['kernel_ms'], d['ess_wall_s']['adaptation']*1e3, d['mean_leapfrogs_per_draw']))"
done
