#!/bin/bash
# radon sampling kernel / warmup, development builds (-DEXMC_DEV_ONLY=2) alternating on one box
out=gpurun_out/$1; n=$2; shift 2; mkdir -p $out
last="${@: -1}"
EXMC_HIP_LIB=$PWD/exmc_amd/lib/$last timeout -k 10 600 python3 -m pytest "tests/test_gpu_full_size.py" -x -q -k "radon" > $out/parity.log 2>&1 || { tail -25 $out/parity.log; exit 1; }
tail -1 $out/parity.log
for i in $(seq 1 $n); do
  for lib in "$@"; do
    v=$(basename $lib .so)
    EXMC_HIP_LIB=$PWD/exmc_amd/lib/$lib python3 bench.py --model radon --no-cpu --no-multi-step > $out/$v.$i.json 2> $out/$v.$i.err || { tail -3 $out/$v.$i.err; exit 1; }
    python3 -c "import json; d=json.load(open('$out/$v.$i.json')); print('$v run $i: %.4e lf/s kernel %.2f ms adapt %.1f ms lf %d eps %.17g' % (d['value'], d['roofline']['kernel_ms'], 1e3*d['ess_wall_s']['adaptation'], d['roofline']['leapfrogs_per_launch'], d['step_size']))"
  done
done
