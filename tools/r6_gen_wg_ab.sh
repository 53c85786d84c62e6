#!/bin/bash
# round 6: the generated 500 x 20 logistic regression with the workgroup form of its sampling kernel (eight wavefronts
# around one LDS image of its tables, EXMC_GEN_WG) against the one-wave form (EXMC_HIP_NUTS_WG=0), alternating on one
# box; the lane layouts' bit-exactness tests first.   gpurun -- 'bash tools/r6_gen_wg_ab.sh'
out=gpurun_out/r6_gen_wg; mkdir -p $out
python3 -m pytest tests/test_gpu_codegen_lanes.py tests/test_gpu_codegen_fast_window.py -x -q > $out/parity.log 2>&1 || { tail -20 $out/parity.log; exit 1; }
tail -1 $out/parity.log
for i in 1 2 3; do for v in wg one; do
  if [ $v = one ]; then export EXMC_HIP_NUTS_WG=0; else unset EXMC_HIP_NUTS_WG; fi
  python3 bench.py --model gen_logistic --no-cpu --no-multi-step > $out/gen_logistic.$v.$i.json 2> $out/gen_logistic.$v.$i.err || { tail -3 $out/gen_logistic.$v.$i.err; exit 1; }
  python3 -c "import json; d=json.load(open('$out/gen_logistic.$v.$i.json')); print('gen_logistic $v: kernel %.1f ms adapt %.4f s eps %.17g lf %d lf/s %.4e ess/s %.4e' % (d['roofline']['kernel_ms'], d['ess_wall_s']['adaptation'], d['step_size'], d['roofline']['leapfrogs_per_launch'], d['value'], d['ess_per_s']))"
done; done
for m in gen_sv gen_radon; do
  unset EXMC_HIP_NUTS_WG
  python3 bench.py --model $m --no-cpu --no-multi-step > $out/$m.json 2> $out/$m.err || { tail -3 $out/$m.err; exit 1; }
  python3 -c "import json; d=json.load(open('$out/$m.json')); print('$m: kernel %.1f ms adapt %.4f s eps %.17g lf %d lf/s %.4e' % (d['roofline']['kernel_ms'], d['ess_wall_s']['adaptation'], d['step_size'], d['roofline']['leapfrogs_per_launch'], d['value']))"
done
