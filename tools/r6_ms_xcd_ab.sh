#!/bin/bash
# Round 6: the batched-leapfrog kernel (B2, multi_step_kernel) with the XCD-contiguous block mapping against the identity
# mapping, alternating on one box: parity of the kernel first, then `roofline_multi_step` of short bench runs.
#   gpurun --timeout 900 -- 'bash tools/r6_ms_xcd_ab.sh <lib before> [rounds]'
before=$1; n=${2:-5}; out=gpurun_out/r6_ms_xcd; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_window_positions.py -m gpu -q -k "multi_step" > $out/parity.log 2>&1 || { tail -5 $out/parity.log; exit 1; }
tail -1 $out/parity.log
for i in $(seq 1 $n); do
  for v in before after; do
    lib=""; [ $v = before ] && lib=$before
    EXMC_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --no-sv-leg --no-extra-legs --no-cpu --steps 2 --warmup 1 > $out/$v.$i.json 2> $out/$v.$i.err || { tail -3 $out/$v.$i.err; exit 1; }
  done
done
python3 - <<PY
import json, glob
for v in ("before", "after"):
    r = [json.load(open(f))["roofline_multi_step"] for f in sorted(glob.glob("$out/%s.*.json" % v))]
    print(v, " ".join("%.3f" % x["frac"] for x in r), " ms:", " ".join("%.4f" % x["kernel_ms"] for x in r))
PY
