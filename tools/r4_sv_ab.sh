#!/bin/bash
# sv sampling kernel A/B on one box: the shipped library against sv-only development builds
# (exmc_amd/lib/libexmc_hip_sv*.so, -DEXMC_DEV_ONLY=1), parity first.
#   gpurun -- 'bash tools/r4_sv_ab.sh <tag> [pmc] lib1.so lib2.so ...'
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
pmc=0; [ "$1" = pmc ] && { pmc=1; shift; }
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2: %.4e lf/s kernel %.1f ms adapt %.3f s eps %.17g lf %d rhat %.6f agree %s' % (d['value'], d['roofline']['kernel_ms'], d['ess_wall_s']['adaptation'], d['step_size'], d['roofline']['leapfrogs_per_launch'], d['rhat_max'], d.get('rhat_routes_agree')))"; }
python3 bench.py --model sv --no-cpu --no-multi-step > $out/base.json 2> $out/base.err || { tail -5 $out/base.err; exit 1; }
line $out/base.json shipped
for lib in "$@"; do
  n=$(basename $lib .so)
  EXMC_HIP_LIB=$PWD/$lib timeout -k 10 600 python3 -m pytest "tests/test_gpu_parity.py::test_random_init_and_momentum_in_flat_order_bit_exact[sv--]" "tests/test_gpu_parity.py::test_bench_protocol_other_models_bit_exact[sv----]" "tests/test_gpu_full_size.py::test_every_chain_of_a_batch_bit_exact[sv-64-12-12]" "tests/test_gpu_full_size.py::test_other_baseline_configs_at_full_size[sv----]" tests/test_gpu_parity.py::test_chain_migration_bit_exact -x -q > $out/parity_$n.log 2>&1 || { tail -25 $out/parity_$n.log; exit 1; }
  tail -1 $out/parity_$n.log
  for i in 1 2; do
    EXMC_HIP_LIB=$PWD/$lib python3 bench.py --model sv --no-cpu --no-multi-step > $out/$n.$i.json 2> $out/$n.$i.err || { tail -5 $out/$n.$i.err; exit 1; }
    line $out/$n.$i.json $n
  done
  if [ $pmc = 1 ]; then
    EXMC_HIP_LIB=$PWD/$lib rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_$n -o run -- python3 bench.py --model sv --no-cpu --no-multi-step > $out/pmc_$n.json 2> $out/pmc_$n.err || { tail -3 $out/pmc_$n.err; exit 1; }
    python3 tools/pmc_kernel_table.py $out/pmc_$n nuts_kernel | tail -2
  fi
done
python3 bench.py --model sv --no-cpu --no-multi-step > $out/base2.json 2> $out/base2.err && line $out/base2.json shipped-again
