#!/bin/bash
# logistic: the matrix-core layout (4 lanes per chain, 16 chains per wavefront) against the vector
# layout (16 lanes per chain) in the SAMPLING kernel, at BASELINE's 8192 chains and at a batch that
# gives the matrix layout a wave per SIMD slot (32768 chains = 2048 waves)
out=gpurun_out/${1:-mfma}; mkdir -p $out
for cfg in "4 8192" "16 8192" "4 32768" "16 32768"; do
  set -- $cfg
  python bench.py --model logistic --lanes $1 --warmup-lanes 64 --chains-per-gpu $2 --no-cpu > $out/bench_l$1_c$2.json 2> $out/bench_l$1_c$2.err || { tail -3 $out/bench_l$1_c$2.err; exit 1; }
  python -c "import json; d=json.load(open('$out/bench_l$1_c$2.json')); print('lanes $1 chains $2: %.3e lf/s kernel %.1f ms' % (d['value'], d['roofline']['kernel_ms']))"
done
