#!/bin/bash
# SQ instruction-class counters of the timed nuts_kernel launch for a radon development build:
#   gpurun -- 'bash tools/r5_rd_pmc.sh <tag> <lib.so>'
tag=$1; lib=$2; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
export EXMC_HIP_LIB=$PWD/$lib
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -o run -- python3 bench.py --model radon --no-cpu --no-multi-step > $out/$name.json 2> $out/$name.err || { tail -3 $out/$name.err; exit 1; }; echo "$name done"; }
run insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS
run insts2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT
run cycles SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
run cyc2 SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
for p in insts insts2 cycles cyc2; do python3 tools/pmc_kernel_table.py $out/$p nuts_kernel | tail -1; done
