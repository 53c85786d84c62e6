#!/bin/bash
# Round 6: the multi-rank control flow of bench.py EXECUTED on the one-GPU box -- the driver's own launch line
# (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
# --gpus N --steps K --warmup W) with EXMC_BENCH_SHARED_GPU=1: N ranks on cuda:0, gloo between them (RCCL refuses two
# ranks on one device; its branches run with one rank in tests/test_gpu_rccl_one_rank.py). Not a measurement: what is
# checked is that N ranks start, rendezvous, make the same collectives, that exactly ONE JSON line reaches stdout and
# that its chain-count-independent figures (step size, R-hat of the first shard's seeds ...) are sane.
#   gpurun --timeout 900 -- 'bash tools/r6_rehearse_ranks.sh r6_rehearsal'
tag=${1:-r6_rehearsal}; out=gpurun_out/$tag; mkdir -p $out
export EXMC_BENCH_SHARED_GPU=1
for n in 2 3; do
  timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
      bench.py --gpus $n --steps 20 --warmup 5 > $out/launcher_n$n.json 2> $out/launcher_n$n.err || { echo "n=$n failed"; tail -20 $out/launcher_n$n.err; exit 1; }
  echo "launcher n=$n: $(wc -l < $out/launcher_n$n.json) line(s) on stdout"
done
# without a launcher: bench.py starts its own ranks (spawn_ranks)
timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-extra-legs > $out/spawn_n2.json 2> $out/spawn_n2.err || { echo "spawn failed"; tail -20 $out/spawn_n2.err; exit 1; }
echo "spawn n=2: $(wc -l < $out/spawn_n2.json) line(s) on stdout"
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$out/*.json")):
    txt = open(f).read().strip().splitlines()
    assert len(txt) == 1, (f, len(txt))
    d = json.loads(txt[0])
    legs = [d] + list(d.get("models", {}).values())
    print(f, d["n_gpus"], d.get("rehearsal", "")[:40])
    for v in legs:
        print("  %-16s chains %s  lf/s %.3e  eps %.6f  rhat %.4f agree %s  ess/s %.3e  gather %s" % (
            v["config"]["workload"][:16], v["config"]["workload"].split("(")[1].split(" ")[0], v["value"], v["step_size"],
            v["rhat_max"], v["rhat_routes_agree"], v["ess_per_s"], v["gather"]["counted"]))
PY
