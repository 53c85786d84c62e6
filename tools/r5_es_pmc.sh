#!/bin/bash
# what eight_schools' lone waves wait for: SQ wait / instruction-fetch counters of the timed nuts_kernel launch
#   gpurun -- 'bash tools/r5_es_pmc.sh <tag>'
tag=$1; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -o run -- python3 bench.py --no-sv-leg --no-extra-legs --no-cpu --no-multi-step > $out/$name.json 2> $out/$name.err || { tail -3 $out/$name.err; return 1; }; python3 tools/pmc_kernel_table.py $out/$name nuts_kernel | tail -1 | cut -c100-; }
run w1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run w2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
run w3 SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INSTS_FLAT
run w4 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run w5 SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC
