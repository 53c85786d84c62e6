#!/bin/bash
# per-wave spans of the logistic sampling launch (libexmc_hip_lgprobe.so: -DEXMC_DEV_ONLY=3 -DEXMC_XCC_PROBE)
out=gpurun_out/${1:-lgprobe}; mkdir -p $out
export EXMC_HIP_LIB=$PWD/exmc_amd/lib/libexmc_hip_lgprobe.so
for prio in 0 1; do
  EXMC_HIP_PRIO=$prio EXMC_WAVE_PROBE_OUT=$out/waves_prio$prio.txt python bench.py --model logistic --no-cpu --no-multi-step > $out/bench_prio$prio.json 2> $out/bench_prio$prio.err || { tail -3 $out/bench_prio$prio.err; exit 1; }
  python -c "import json; d=json.load(open('$out/bench_prio$prio.json')); print('prio $prio: %.3e lf/s kernel %.1f ms' % (d['value'], d['roofline']['kernel_ms']))"
done
