#!/bin/bash
# sv at the protocol (2048 chains x 1000 draws): the bench line with its CPU leg, then the same
# command under rocprofv3 for the kernel summary. Output under gpurun_out/$1/.
set -o pipefail
tag=${1:-sv_session}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python bench.py --model sv > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
cat $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o run -- python3 bench.py --model sv --no-cpu > $out/bench_under_rocprof.json 2> $out/rocprof.err || { tail -5 $out/rocprof.err; exit 1; }
find $out/prof -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
head -6 $out/kernel_stats.csv
