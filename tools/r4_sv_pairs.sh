#!/bin/bash
# Is the slow state of the sv launch a mis-paired priority schedule? Probe build (-DEXMC_XCC_PROBE): every workgroup
# records the SIMD it ran on; the time-sliced priority assumes workgroups b and b + 1024 share one.
out=gpurun_out/${1:-r4_sv_pairs}; mkdir -p $out; n=${2:-8}
export EXMC_HIP_LIB=$PWD/exmc_amd/lib/libexmc_hip_svprobe.so
for i in $(seq 1 $n); do
  EXMC_WAVE_PROBE_OUT=$out/waves$i.txt python3 tools/r4_sv_place.py ${3:-none} > $out/run$i.txt 2>/dev/null
  python3 - <<PY
import numpy as np
f = open("$out/waves$i.txt"); ms = float(f.readline()); a = np.loadtxt(f)
place = a[:2048, 1]
same = int((place[:1024] == place[1024:2048]).sum())
u, c = np.unique(place, return_counts=True)
print("run $i: kernel %.1f ms; pairs (b, b+1024) on one SIMD: %d of 1024; SIMDs used %d, with exactly two workgroups %d, max on one SIMD %d" % (ms, same, len(u), int((c == 2).sum()), int(c.max())))
PY
done
