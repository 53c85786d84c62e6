#!/bin/bash
# eight_schools sampling kernel as wave pairs (tree + integrator, EXMC_HIP_NUTS_PIPE=1) against the
# one-wave form, with and without the role swap + tree-wave priority (EXMC_HIP_PRIO), on the probe
# build (libexmc_hip_esprobe.so: -DEXMC_DEV_ES16_ONLY -DEXMC_XCC_PROBE): bench lines and, for the
# pair form, which waves share a SIMD.
#   gpurun -- 'bash tools/es_pipe_probe.sh r3_espipe'
out=gpurun_out/${1:-espipe}; mkdir -p $out
export EXMC_HIP_LIB=${EXMC_HIP_LIB:-$PWD/exmc_amd/lib/libexmc_hip_esprobe.so}
for cfg in "0 1" "1 0" "1 1"; do
  set -- $cfg; pipe=$1; prio=$2
  EXMC_HIP_NUTS_PIPE=$pipe EXMC_HIP_PRIO=$prio EXMC_WAVE_PROBE_OUT=$out/waves_pipe${pipe}_prio$prio.txt python bench.py --no-cpu --no-sv-leg --no-multi-step > $out/bench_pipe${pipe}_prio$prio.json 2> $out/bench_pipe${pipe}_prio$prio.err || { tail -3 $out/bench_pipe${pipe}_prio$prio.err; exit 1; }
  python -c "import json; d=json.load(open('$out/bench_pipe${pipe}_prio$prio.json')); print('pipe $pipe prio $prio: %.3e lf/s kernel %.2f ms eps %.6f lf %d' % (d['value'], d['roofline']['kernel_ms'], d['step_size'], d['roofline']['leapfrogs_per_launch']))"
done
python - <<PY
import collections
for name in ("waves_pipe1_prio0.txt", "waves_pipe1_prio1.txt"):
    rows = [l.split() for l in open("$out/" + name).read().splitlines()[1:]]
    simd = collections.defaultdict(list)
    for r in rows[:2048]:
        e, place, role = int(r[0]), int(float(r[1])), int(float(r[2]))
        if place:
            simd[place].append((e // 2, e % 2, role))
    kinds = collections.Counter(tuple(sorted(x[2] for x in v)) for v in simd.values())
    print(name, "SIMDs", len(simd), "roles per SIMD (0 tree, 1 integrator):", dict(kinds))
    print("  examples:", [(k, v) for k, v in list(simd.items())[:6]])
PY
