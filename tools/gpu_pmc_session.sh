#!/bin/bash
# PMC passes for the timed nuts_kernel launch of `python3 bench.py --no-cpu` (separate passes, each
# with --kernel-trace only, as MI355X_MICROARCH.md prescribes):
#   gpurun --timeout 900 -- 'bash tools/gpu_pmc_session.sh <tag> [bench args...]'
# then: python tools/pmc_summary.py gpurun_out/<tag> profiles/<name>
tag=${1:-pmc}; shift; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -o run -- python3 bench.py --no-cpu $BENCH_ARGS > $out/$name.json 2> $out/$name.err || { tail -3 $out/$name.err; exit 1; }; echo "$name done"; }
BENCH_ARGS="$*"
run fetch FETCH_SIZE
run write WRITE_SIZE
run insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS
run cycles SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
run f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 bench.py --no-cpu $BENCH_ARGS > $out/bench_under_rocprof.json 2> $out/stats.err
python bench.py $BENCH_ARGS > $out/bench.json 2> $out/bench.err
ls $out
