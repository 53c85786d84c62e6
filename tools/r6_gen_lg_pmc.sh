#!/bin/bash
# what the generated logistic regression's sampling kernel waits for, next to the hand-written kind's: four counter
# passes each (--kernel-trace only).   gpurun --timeout 900 -- 'bash tools/r6_gen_lg_pmc.sh'
out=gpurun_out/r6_gen_lg_pmc; mkdir -p $out
export TMPDIR=/tmp
for m in gen_logistic logistic; do
  run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$m.$name -o run -- python3 bench.py --no-cpu --no-multi-step --model $m > $out/$m.$name.json 2> $out/$m.$name.err || { tail -3 $out/$m.$name.err; exit 1; }; echo "$m $name done"; }
  run insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS
  run cycles SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
  run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
  run mem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU
done
python3 - <<'PY'
import csv, glob, collections, os
out = "gpurun_out/r6_gen_lg_pmc"
for m in ("gen_logistic", "logistic"):
    tot = collections.OrderedDict()
    for d in sorted(glob.glob(out + "/%s.*/" % m)):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            rows = list(csv.DictReader(open(f)))
            # the timed launch = the nuts_kernel dispatch with the largest grid (the batch), last one
            ks = [r for r in rows if "nuts_kernel" in r["Kernel_Name"]]
            if not ks:
                continue
            big = max(int(r["Grid_Size"]) for r in ks)
            last = max(int(r["Dispatch_Id"]) for r in ks if int(r["Grid_Size"]) == big)
            for r in ks:
                if int(r["Dispatch_Id"]) == last:
                    tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    print(m, dict(tot))
PY
