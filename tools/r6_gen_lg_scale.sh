out=gpurun_out/r6_gen_lg_scale; mkdir -p $out
for m in gen_logistic logistic; do for c in 1024 2048 4096 8192 16384; do
  python3 bench.py --model $m --chains-per-gpu $c --no-cpu --no-multi-step > $out/$m.$c.json 2> $out/$m.$c.err || { tail -3 $out/$m.$c.err; exit 1; }
  python3 -c "import json; d=json.load(open('$out/$m.$c.json')); print('$m chains $c: kernel %.1f ms lf %d lf/s %.4e' % (d['roofline']['kernel_ms'], d['roofline']['leapfrogs_per_launch'], d['value']))"
done; done
