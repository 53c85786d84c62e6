#!/bin/bash
# One GPU-box session: GPU tests, the bench line, a rocprofv3 kernel summary of the same command and
# (optionally) development probes. Everything lands under gpurun_out/$1/.
#   gpurun --timeout 900 -- 'bash tools/gpu_session.sh r2a [probe] [sections]'
set -o pipefail
tag=${1:-session}; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
python bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
cat $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o run -- python3 bench.py --no-cpu > $out/bench_under_rocprof.json 2> $out/rocprof.err || { tail -5 $out/rocprof.err; exit 1; }
find $out/prof -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
head -8 $out/kernel_stats.csv
for what in "$@"; do
  case $what in
    probe) ./tools/probe/exec_mask_rate_probe > $out/exec_mask_rate_probe.txt 2>&1; cat $out/exec_mask_rate_probe.txt ;;
    fmac) ./tools/probe/fmac_dpp_probe > $out/fmac_dpp_probe.txt 2>&1; cat $out/fmac_dpp_probe.txt ;;
    sections) EXMC_HIP_LIB=$PWD/exmc_amd/lib/libexmc_hip_prof.so python bench.py --no-cpu > $out/bench_sections.json 2> $out/sections.txt; grep "exmc prof" $out/sections.txt | tail -12 ;;
  esac
done
