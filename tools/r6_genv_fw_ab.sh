#!/bin/bash
# round 6: the generated PLATE layout (16 lanes per chain, codegen_vec.py) with and without the fast window of its
# exp / log calls (-DEXMC_GEN_FAST_WINDOW=0 through EXMC_GEN_EXTRA_FLAGS: a plug-in of its own cache tag), alternating
# on one box; the bit-exactness tests of the generated layouts first.   gpurun -- 'bash tools/r6_genv_fw_ab.sh'
out=gpurun_out/r6_genv_fw; mkdir -p $out
python3 -m pytest tests/test_gpu_codegen.py tests/test_gpu_codegen_random.py tests/test_gpu_codegen_round2.py \
  tests/test_gpu_codegen_inline.py -x -q > $out/parity.log 2>&1 || { tail -20 $out/parity.log; exit 1; }
tail -1 $out/parity.log
for i in 1 2 3; do for v in fast exact; do
  m=gen_eight_schools
  if [ $v = exact ]; then export EXMC_GEN_EXTRA_FLAGS="-DEXMC_GEN_FAST_WINDOW=0"; else unset EXMC_GEN_EXTRA_FLAGS; fi
  python3 bench.py --model $m --no-cpu --no-multi-step > $out/$m.$v.$i.json 2> $out/$m.$v.$i.err || { tail -3 $out/$m.$v.$i.err; exit 1; }
  python3 -c "import json; d=json.load(open('$out/$m.$v.$i.json')); print('$m $v: kernel %.2f ms adapt %.4f s eps %.17g lf %d' % (d['roofline']['kernel_ms'], d['ess_wall_s']['adaptation'], d['step_size'], d['roofline']['leapfrogs_per_launch']))"
done; done
