#!/bin/bash
# logistic sampling kernel on one box: parity of the tree's library with the workgroup form of the
# sampling kernel forced on (EXMC_HIP_NUTS_WG=1: small batches would take the one-wave form), forced
# off and as dispatched; then bench lines alternating with reference libraries (e.g. the round-4
# library kept as exmc_amd/lib/libexmc_hip_r4.so), optionally one SQ counter pass.
#   gpurun -- 'bash tools/r5_lg_ab.sh <tag> [pmc] [other.so ...]'
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
pmc=0; [ "$1" = pmc ] && { pmc=1; shift; }
line() { python3 -c "import json,sys; d=json.load(open('$1')); print('$2: %.4e lf/s kernel %.1f ms adapt %.3f s eps %.17g lf %d ess/s %.3e rhat %.6f' % (d['value'], d['roofline']['kernel_ms'], d['ess_wall_s']['adaptation'], d['step_size'], d['roofline']['leapfrogs_per_launch'], d['ess_per_s'], d['rhat_max']))"; }
for wg in 1 0 x; do
  if [ $wg = x ]; then unset EXMC_HIP_NUTS_WG; else export EXMC_HIP_NUTS_WG=$wg; fi
  timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -k logistic -x -q > $out/parity_wg$wg.log 2>&1 || { tail -30 $out/parity_wg$wg.log; exit 1; }
  echo "wg=$wg: $(tail -1 $out/parity_wg$wg.log)"
done
unset EXMC_HIP_NUTS_WG
for i in 1 2; do
  python3 bench.py --model logistic --no-cpu --no-multi-step > $out/tree.$i.json 2> $out/tree.$i.err || { tail -5 $out/tree.$i.err; exit 1; }
  line $out/tree.$i.json tree
  for lib in "$@"; do
    n=$(basename $lib .so)
    EXMC_HIP_LIB=$PWD/$lib python3 bench.py --model logistic --no-cpu --no-multi-step > $out/$n.$i.json 2> $out/$n.$i.err || { tail -5 $out/$n.$i.err; exit 1; }
    line $out/$n.$i.json $n
  done
done
EXMC_HIP_NUTS_WG=0 python3 bench.py --model logistic --no-cpu --no-multi-step > $out/tree.onewave.json 2> $out/tree.onewave.err || { tail -5 $out/tree.onewave.err; exit 1; }
line $out/tree.onewave.json "tree, one-wave form"
if [ $pmc = 1 ]; then
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_tree -o run -- python3 bench.py --model logistic --no-cpu --no-multi-step > $out/pmc_tree.json 2> $out/pmc_tree.err || { tail -3 $out/pmc_tree.err; exit 1; }
  python3 tools/pmc_kernel_table.py $out/pmc_tree nuts_kernel | tail -2
fi
