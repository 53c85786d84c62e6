#!/bin/bash
# where does the 1592 / 1645 / 1690 ms spread of the sv leg in the driver's command come from?
out=gpurun_out/${1:-r4_sv_var}; mkdir -p $out
show() { python3 -c "
import json,sys
d=json.load(open('$1'))
sv=d['models']['sv'] if 'models' in d else d
print('$2: sv kernel %.1f ms' % sv['roofline']['kernel_ms'])"; }
for i in 1 2 3; do
  python3 bench.py --model sv --no-cpu --no-multi-step --warmup 5 > $out/a$i.json 2>/dev/null && show $out/a$i.json "sv alone W=5"
  python3 bench.py --model sv --no-cpu --no-multi-step --warmup 2 > $out/b$i.json 2>/dev/null && show $out/b$i.json "sv alone W=2"
  python3 bench.py --no-cpu --no-multi-step --warmup 2 > $out/c$i.json 2>/dev/null && show $out/c$i.json "es+sv W=2 no cpu leg"
  python3 bench.py --no-cpu --no-multi-step --warmup 5 > $out/d$i.json 2>/dev/null && show $out/d$i.json "es+sv W=5 no cpu leg"
  python3 bench.py --warmup 5 > $out/e$i.json 2>/dev/null && show $out/e$i.json "driver command"
done
