#!/bin/bash
# sv sampling kernel, round 3 against round 4, alternating on one box (sv-only development builds of
# the two trees: exmc_amd/lib/libexmc_hip_sv_r3.so from commit 840744f, libexmc_hip_sv_r4.so from the
# working tree). Kernel time of the 2048 x 1000 launch from the library's HIP events.
out=gpurun_out/${1:-r4_sv_ab2}; mkdir -p $out; n=${2:-4}
for i in $(seq 1 $n); do
  for v in r3 r4; do
    EXMC_HIP_LIB=$PWD/exmc_amd/lib/libexmc_hip_sv_$v.so python3 bench.py --model sv --no-cpu --no-multi-step > $out/$v.$i.json 2> $out/$v.$i.err || { tail -3 $out/$v.$i.err; exit 1; }
    python3 -c "import json; d=json.load(open('$out/$v.$i.json')); print('$v run $i: %.4e lf/s kernel %.1f ms adapt %.3f s lf %d eps %.17g' % (d['value'], d['roofline']['kernel_ms'], d['ess_wall_s']['adaptation'], d['roofline']['leapfrogs_per_launch'], d['step_size']))"
  done
done
