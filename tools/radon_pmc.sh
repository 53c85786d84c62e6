#!/bin/bash
out=gpurun_out/r3radpmc; mkdir -p $out; export TMPDIR=/tmp
for m in radon gen_radon; do
  EXTRA=""; [ $m = radon ] && [ -f exmc_amd/lib/libexmc_hip_radon.so ] && export EXMC_HIP_LIB=$PWD/exmc_amd/lib/libexmc_hip_radon.so || unset EXMC_HIP_LIB
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $out/$m -o run -- python3 bench.py --model $m --no-cpu --no-multi-step > $out/$m.json 2> $out/$m.err || { tail -3 $out/$m.err; exit 1; }
  python tools/pmc_kernel_table.py $out/$m nuts_kernel | tail -1
done
