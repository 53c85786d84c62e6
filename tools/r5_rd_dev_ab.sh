#!/bin/bash
# radon-only development builds (exmc_amd/lib/libexmc_hip_rd*.so, -DEXMC_DEV_ONLY=2) alternating on one box:
# the parity tests that need only the 64-lane layout first (for the LAST library named), then bench lines
# (kernel ms, adaptation s, step size and leapfrog count -- equal bits show there too).
#   gpurun -- 'bash tools/r5_rd_dev_ab.sh <tag> lib1.so lib2.so ...'
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2: %.4e lf/s kernel %.2f ms adapt %.4f s ess/s %.4e eps %.17g lf %d' % (d['value'], d['roofline']['kernel_ms'], d['ess_wall_s']['adaptation'], d['ess_per_s'], d['step_size'], d['roofline']['leapfrogs_per_launch']))"; }
last="${@: -1}"
EXMC_HIP_LIB=$PWD/$last timeout -k 10 600 python3 -m pytest "tests/test_golden_traces.py::test_hip_reproduces_committed_traces[radon_g64]" "tests/test_gpu_full_size.py::test_every_chain_of_a_batch_bit_exact[radon-64-48-40]" "tests/test_gpu_full_size.py::test_other_baseline_configs_at_full_size[radon----]" "tests/test_gpu_parity.py::test_bench_protocol_other_models_bit_exact[radon----]" "tests/test_gpu_parity.py::test_logp_grad_extreme_operands_bit_exact[radon--]" "tests/test_gpu_parity.py::test_random_init_and_momentum_in_flat_order_bit_exact[radon--]" -x -q > $out/parity.log 2>&1 || { tail -25 $out/parity.log; exit 1; }
echo "$(basename $last .so): $(tail -1 $out/parity.log)"
for i in 1 2 3; do
  for lib in "$@"; do
    n=$(basename $lib .so)
    EXMC_HIP_LIB=$PWD/$lib python3 bench.py --model radon --no-cpu --no-multi-step > $out/$n.$i.json 2> $out/$n.$i.err || { tail -5 $out/$n.$i.err; exit 1; }
    line $out/$n.$i.json $n
  done
done
