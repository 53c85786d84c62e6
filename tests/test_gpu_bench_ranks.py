"""bench.py's multi-rank control flow EXECUTED on the one-GPU box: the driver's own launch line
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N
...`) with EXMC_BENCH_SHARED_GPU=1 -- N ranks on cuda:0, gloo between them (RCCL refuses two ranks on one device; its
branches run with one rank in tests/test_gpu_rccl_one_rank.py). Chains shard in contiguous blocks and chain i keeps seed
base + 7919 i, so two ranks of 256 chains ARE one rank of 512: every deterministic figure of the line must
be the one-rank line's."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--steps", "4", "--warmup", "1", "--no-cpu", "--no-multi-step", "--no-extra-legs"]


def _port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _line(cmd, env):
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = r.stdout.decode().strip().splitlines()
    assert len(out) == 1, out            # ONE JSON line, whatever the ranks and their libraries print
    return json.loads(out[0])


def test_two_ranks_are_one_rank_of_twice_the_chains(hip):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    bench = os.path.join(ROOT, "bench.py")
    one = _line([sys.executable, bench, "--gpus", "1", "--chains-per-gpu", "512"] + ARGS, env)
    two = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                 "127.0.0.1", "--master-port", str(_port()), bench, "--gpus", "2", "--chains-per-gpu", "256"] + ARGS,
                dict(env, EXMC_BENCH_SHARED_GPU="1"))
    assert two["n_gpus"] == 2 and "rehearsal" in two and "rehearsal" not in one
    assert "(512 total)" in one["config"]["workload"] and "(512 total)" in two["config"]["workload"]
    for k in ("step_size", "divergent_transitions", "mean_leapfrogs_per_draw"):
        assert two[k] == one[k], k
    for k in ("ess_min_total", "ess_bulk_min_total", "rhat_max", "rhat_max_from_chain_stats"):
        assert abs(two[k] - one[k]) <= 1e-12 * abs(one[k]), k      # sums over ranks associate differently
    assert two["rhat_routes_agree"] is True and one["rhat_routes_agree"] is True
    # the sv leg rides on the line with its trace all-gather (2048 chains per rank: 4096 against 2048, other chains,
    # same tuning -- the warmup is chain 0's on every rank)
    sv1, sv2 = one["models"]["sv"], two["models"]["sv"]
    assert sv2["gather"]["counted"] == "traces" and sv2["gather"]["bytes_per_rank"] == sv1["gather"]["bytes_per_rank"]
    assert sv2["step_size"] == sv1["step_size"] and sv2["rhat_routes_agree"] is True
    assert "(4096 total)" in sv2["config"]["workload"]
