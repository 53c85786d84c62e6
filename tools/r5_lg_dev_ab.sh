#!/bin/bash
# logistic-only development builds (exmc_amd/lib/libexmc_hip_lg*.so, -DEXMC_DEV_ONLY=3) alternating on one
# box: the parity tests that need only the 16- and 64-lane layouts first, then bench lines.
#   gpurun -- 'bash tools/r5_lg_dev_ab.sh <tag> lib1.so lib2.so ...'
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2: %.4e lf/s kernel %.1f ms adapt %.4f s eps %.17g lf %d' % (d['value'], d['roofline']['kernel_ms'], d['ess_wall_s']['adaptation'], d['step_size'], d['roofline']['leapfrogs_per_launch']))"; }
for lib in "$@"; do
  n=$(basename $lib .so)
  EXMC_HIP_LIB=$PWD/$lib timeout -k 10 600 python3 -m pytest "tests/test_gpu_parity.py::test_bench_protocol_other_models_bit_exact[logistic----]" "tests/test_gpu_full_size.py::test_every_chain_of_a_batch_bit_exact[logistic-16-96-40]" "tests/test_gpu_full_size.py::test_other_baseline_configs_at_full_size[logistic----]" tests/test_gpu_parity.py::test_logistic_warmup_layout_differs_from_sampling_layout -x -q > $out/parity_$n.log 2>&1 || { tail -25 $out/parity_$n.log; exit 1; }
  echo "$n: $(tail -1 $out/parity_$n.log)"
done
for i in 1 2 3; do
  for lib in "$@"; do
    n=$(basename $lib .so)
    EXMC_HIP_LIB=$PWD/$lib python3 bench.py --model logistic --no-cpu --no-multi-step > $out/$n.$i.json 2> $out/$n.$i.err || { tail -5 $out/$n.$i.err; exit 1; }
    line $out/$n.$i.json $n
  done
done
