#!/bin/bash
# cross-row stages of the 64-lane sums: LDS crossbar (pl0) against v_permlane16/32_swap + DPP (pl1)
out=gpurun_out/${1:-r4_sv_pl}; mkdir -p $out
python3 -m pytest tests/test_gpu_allsum_rs.py -x -q 2>&1 | tail -2
bash tools/r4_sv_ab3.sh ${1:-r4_sv_pl} 3 libexmc_hip_sv_pl0.so libexmc_hip_sv_pl1.so
for i in 1 2; do for v in 0 1; do
  EXMC_HIP_LIB=$PWD/exmc_amd/lib/libexmc_hip_sv_pl$v.so python3 bench.py --model sv --chains-per-gpu 1024 --steps 4 --no-cpu --no-multi-step > $out/lone$v.json 2> $out/lone$v.err || { tail -3 $out/lone$v.err; exit 1; }
  python3 -c "import json; d=json.load(open('$out/lone$v.json')); print('permlane=$v lone waves (1024 x 200): kernel %.1f ms' % d['roofline']['kernel_ms'])"
done; done
