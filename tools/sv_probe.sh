#!/bin/bash
# per-wave placement / span / leapfrogs of the sv sampling launch (libexmc_hip_svprobe.so: -DEXMC_DEV_ONLY=1
# -DEXMC_XCC_PROBE), time-sliced priority off / on, chain migration off / on:
#   gpurun -- 'bash tools/sv_probe.sh r3_svprobe'
out=gpurun_out/${1:-svprobe}; mkdir -p $out
export EXMC_HIP_LIB=$PWD/exmc_amd/lib/libexmc_hip_svprobe.so
for cfg in "0 0" "0 1" "1 0" "1 1"; do
  set -- $cfg; prio=$1; mig=$2
  EXMC_HIP_PRIO=$prio EXMC_HIP_MIGRATE=$mig EXMC_WAVE_PROBE_OUT=$out/waves_prio${prio}_mig$mig.txt python bench.py --model sv --no-cpu --no-multi-step > $out/bench_prio${prio}_mig$mig.json 2> $out/bench_prio${prio}_mig$mig.err || { tail -3 $out/bench_prio${prio}_mig$mig.err; exit 1; }
  python -c "import json; d=json.load(open('$out/bench_prio${prio}_mig$mig.json')); print('prio $prio migrate $mig: %.3e lf/s kernel %.1f ms eps %.5f' % (d['value'], d['roofline']['kernel_ms'], d['step_size']))"
done
