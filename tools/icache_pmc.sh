#!/bin/bash
out=gpurun_out/icache; mkdir -p $out; export TMPDIR=/tmp
for m in sv radon; do
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES --output-format csv -d $out/$m -o run -- python3 bench.py --model $m --no-cpu --no-multi-step > $out/$m.json 2> $out/$m.err || { tail -5 $out/$m.err; exit 1; }
python tools/pmc_kernel_table.py $out/$m nuts_kernel | tail -1
done
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES --output-format csv -d $out/es -o run -- python3 bench.py --no-cpu --no-multi-step --no-sv-leg > $out/es.json 2> $out/es.err || { tail -5 $out/es.err; exit 1; }
python tools/pmc_kernel_table.py $out/es nuts_kernel | tail -1
