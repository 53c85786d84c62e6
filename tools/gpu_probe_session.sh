#!/bin/bash
# development probes on the GPU box: gpurun -- 'bash tools/gpu_probe_session.sh <tag>'
tag=${1:-probe}; out=gpurun_out/$tag; mkdir -p $out
./tools/probe/exec_mask_rate_probe > $out/exec_mask_rate_probe.txt 2>&1; cat $out/exec_mask_rate_probe.txt
python bench.py --no-cpu > $out/bench_default.json 2> $out/err1.txt; python -c "import json;d=json.load(open('$out/bench_default.json'));print('default  kernel_ms',d['roofline']['kernel_ms'],'lf/s',d['value'])"
EXMC_HIP_NUTS_PIPE=1 python bench.py --no-cpu > $out/bench_pipe.json 2> $out/err2.txt; python -c "import json;d=json.load(open('$out/bench_pipe.json'));print('wave-pair kernel_ms',d['roofline']['kernel_ms'],'lf/s',d['value'])"
