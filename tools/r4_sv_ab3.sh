#!/bin/bash
# parity of an sv-only development build, then its kernel time alternating with other builds on one box:
#   gpurun -- 'bash tools/r4_sv_ab3.sh <tag> <n> libA.so libB.so ...'   (libs under exmc_amd/lib/)
out=gpurun_out/$1; n=$2; shift 2; mkdir -p $out
last="${@: -1}"
EXMC_HIP_LIB=$PWD/exmc_amd/lib/$last timeout -k 10 600 python3 -m pytest "tests/test_gpu_parity.py::test_random_init_and_momentum_in_flat_order_bit_exact[sv--]" "tests/test_gpu_parity.py::test_bench_protocol_other_models_bit_exact[sv----]" "tests/test_gpu_full_size.py::test_every_chain_of_a_batch_bit_exact[sv-64-12-12]" "tests/test_gpu_full_size.py::test_other_baseline_configs_at_full_size[sv----]" tests/test_gpu_parity.py::test_chain_migration_bit_exact -x -q > $out/parity.log 2>&1 || { tail -25 $out/parity.log; exit 1; }
tail -1 $out/parity.log
for i in $(seq 1 $n); do
  for lib in "$@"; do
    v=$(basename $lib .so)
    EXMC_HIP_LIB=$PWD/exmc_amd/lib/$lib python3 bench.py --model sv --no-cpu --no-multi-step > $out/$v.$i.json 2> $out/$v.$i.err || { tail -3 $out/$v.$i.err; exit 1; }
    python3 -c "import json; d=json.load(open('$out/$v.$i.json')); print('$v run $i: %.4e lf/s kernel %.1f ms adapt %.3f s lf %d eps %.17g' % (d['value'], d['roofline']['kernel_ms'], d['ess_wall_s']['adaptation'], d['roofline']['leapfrogs_per_launch'], d['step_size']))"
  done
done
