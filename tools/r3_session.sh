#!/bin/bash
# Round-3 GPU-box session: the steps named on the command line, in order, stopping at the first
# failure. Everything lands under gpurun_out/$1/.
#   gpurun --timeout 1200 -- 'bash tools/r3_session.sh r3a lanes stream bench gen_sv'
set -o pipefail
tag=${1:-session}; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for what in "$@"; do
  echo "== $what $(date +%T)"
  case $what in
    lanes) python -m pytest tests/test_gpu_codegen_lanes.py -x -q > $out/pytest_lanes.log 2>&1; rc=$?; tail -5 $out/pytest_lanes.log; if [ $rc -ne 0 ]; then exit $rc; fi ;;
    stream) python -m pytest tests/test_gpu_parity.py -x -q -k "stream" > $out/pytest_stream.log 2>&1; rc=$?; tail -5 $out/pytest_stream.log; if [ $rc -ne 0 ]; then exit $rc; fi ;;
    shard) python -m pytest tests/test_gpu_sharded_api.py -x -q > $out/pytest_shard.log 2>&1; rc=$?; tail -5 $out/pytest_shard.log; if [ $rc -ne 0 ]; then exit $rc; fi ;;
    nif) python -m pytest tests/test_gpu_nif_shim.py -x -q > $out/pytest_nif.log 2>&1; rc=$?; tail -5 $out/pytest_nif.log; if [ $rc -ne 0 ]; then exit $rc; fi ;;
    all) python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -5 $out/pytest.log; if [ $rc -ne 0 ]; then exit $rc; fi ;;
    bench) python bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }; cat $out/bench.json ;;
    prof) rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o run -- python3 bench.py --no-cpu > $out/bench_under_rocprof.json 2> $out/rocprof.err || { tail -5 $out/rocprof.err; exit 1; }
          find $out/prof -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \; ; rm -rf $out/prof; head -8 $out/kernel_stats.csv ;;
    gen_*|sv|radon|logistic) python bench.py --model $what --no-cpu > $out/bench_$what.json 2> $out/bench_$what.err || { tail -5 $out/bench_$what.err; exit 1; }; cat $out/bench_$what.json ;;
    cpu_*) m=${what#cpu_}; python bench.py --model $m > $out/bench_${m}_cpu.json 2> $out/bench_${m}_cpu.err || { tail -5 $out/bench_${m}_cpu.err; exit 1; }; cat $out/bench_${m}_cpu.json ;;
    cost) python tools/model_cost.py $COST_MODELS > $out/model_cost.txt 2> $out/model_cost.err || { tail -5 $out/model_cost.err; exit 1; }; cat $out/model_cost.txt ;;
    costpmc) rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $out/costpmc -o run -- python3 tools/model_cost.py $COST_MODELS > $out/model_cost_pmc.txt 2> $out/costpmc.err || { tail -5 $out/costpmc.err; exit 1; }
          python tools/pmc_kernel_table.py $out/costpmc multi_step_kernel | tee $out/model_cost_pmc_table.txt ;;
    mfma) rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/mfmapmc -o run -- python3 tools/model_cost.py $COST_MODELS > $out/mfma_cost.txt 2> $out/mfmapmc.err || { tail -5 $out/mfmapmc.err; exit 1; }
          cat $out/mfma_cost.txt; python tools/pmc_kernel_table.py $out/mfmapmc multi_step_kernel | tee $out/mfma_pmc_table.txt ;;
    *) echo "unknown step $what"; exit 2 ;;
  esac
done
