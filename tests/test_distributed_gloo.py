"""CPU test of the N > 1 path: world_size-2 gloo. Each rank samples its contiguous chain block
(here with the CPU checker standing in for the kernels, which need a GPU), the traces are
all-gathered in chain order and the diagnostics reduced exactly as bench.py does on RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_chains, S, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from exmc_amd import distributed as xd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = xd.shard_range(n_chains, rank, world)
    m = O.eight_schools()
    t, st = O.sample_chains(m, n_chains, init_q=np.zeros(10), num_warmup=80, num_samples=S, seed=42,
                            chain_lo=lo, chain_hi=hi)
    # device trace layout [S][D][C_local]
    local = torch.from_numpy(np.ascontiguousarray(t["draws"].transpose(1, 2, 0)))
    allt = xd.gather_traces(local, dist)
    L = O.lib()
    ess = np.zeros(10)
    for c in range(hi - lo):
        for i in range(10):
            ess[i] += L.exo_ess(O.dptr(np.ascontiguousarray(t["draws"][c, :, i])), S)
    ess_t = xd.reduce_sum(torch.from_numpy(ess.copy()), dist)
    lf = xd.reduce_sum(torch.tensor([float(st.total_leapfrogs)], dtype=torch.float64), dist)
    tmax = xd.reduce_max(torch.tensor([float(rank + 1)], dtype=torch.float64), dist)
    rhat = xd.split_rhat(allt)
    # the default bench path: all-gather of per-chain half-chain statistics instead of the traces
    hm, hv, hn = xd.half_chain_stats(local)
    rhat_stats = xd.split_rhat_from_stats(xd.gather_chain_stats(hm, dist),
                                          xd.gather_chain_stats(hv, dist), hn)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), all=allt.numpy(), ess=ess_t.numpy(),
             lf=lf.numpy(), tmax=tmax.numpy(), rhat=rhat.numpy(), rhat_stats=rhat_stats.numpy(),
             eps=st.step_size)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    sys.path.insert(0, ROOT)
    from exmc_amd import distributed as xd
    for n, w in [(4096, 8), (10, 3), (7, 7), (5, 8)]:
        rs = [xd.shard_range(n, r, w) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n
        assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in rs]
        assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_sharding_matches_single_process(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from exmc_amd import distributed as xd
    world, n_chains, S = 2, 6, 40
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_chains, S, str(tmp_path)), nprocs=world, join=True)
    m = O.eight_schools()
    t, st = O.sample_chains(m, n_chains, init_q=np.zeros(10), num_warmup=80, num_samples=S, seed=42)
    ref = np.ascontiguousarray(t["draws"].transpose(1, 2, 0))      # [S][D][C]
    L = O.lib()
    ess = np.zeros(10)
    for c in range(n_chains):
        for i in range(10):
            ess[i] += L.exo_ess(O.dptr(np.ascontiguousarray(t["draws"][c, :, i])), S)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert np.array_equal(z["all"], ref)                 # chain order preserved, bit for bit
        assert np.allclose(z["ess"], ess, rtol=1e-12)
        assert z["lf"][0] == st.total_leapfrogs
        assert z["tmax"][0] == world
        assert float(z["eps"]) == st.step_size               # every rank re-derives the same tuning
        assert np.allclose(z["rhat"], xd.split_rhat(torch.from_numpy(ref)).numpy())
        assert np.allclose(z["rhat_stats"], z["rhat"], rtol=1e-12, atol=0.0)
        assert np.all(np.abs(z["rhat"] - 1.0) < 0.5)


@pytest.mark.parametrize("world", [1, 2, 3])
def test_sample_chains_sharded_api_is_independent_of_the_number_of_ranks(world):
    """exmc_amd.distributed.sample_chains_sharded -- what sampler.sample_chains(ir, n, devices=[...])
    calls -- with the CPU checker as the engine: spawned rank processes, the shared warmup repeated
    on every rank, contiguous chain blocks, results concatenated in chain order. The answer equals
    the one-process answer bit for bit for 1, 2 and 3 ranks (5 chains: unequal and empty shards)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from exmc_amd import distributed as xd
    from exmc_amd import models
    spec = models.eight_schools()
    opts = dict(num_warmup=60, num_samples=25, seed=9, init_values=spec.default_init)
    n_chains = 5 if world < 3 else 2          # 2 chains on 3 ranks: one shard is empty
    traces, stats = xd.sample_chains_sharded(spec, n_chains, opts, devices=list(range(world)),
                                             engine="oracle_engine")
    t, st = O.sample_chains(O.model_for(spec), n_chains, init_q=spec.to_unconstrained(spec.default_init),
                            num_warmup=60, num_samples=25, seed=9, cfg=O.Cfg(1, 1))
    raw = stats[0]["extra"]["raw"]
    assert np.array_equal(raw["draws"], t["draws"]) and np.array_equal(raw["tree_depth"], t["tree_depth"])
    assert stats[0]["step_size"] == st.step_size
    assert stats[0]["extra"]["total_leapfrogs"] == st.total_leapfrogs
    assert len(traces) == n_chains and set(traces[0]) == set(spec.var_names)
    assert np.array_equal(traces[n_chains - 1]["tau"], np.exp(t["draws"][n_chains - 1, :, 1]))
    shards = stats[0]["extra"]["shards"]
    assert shards[0][0] == 0 and shards[-1][1] == n_chains and len(shards) == world


def test_sharded_api_refuses_what_it_cannot_send_to_the_ranks():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from exmc_amd import distributed as xd
    from exmc_amd import models
    spec = models.eight_schools()
    spec.closure = lambda x: x            # e.g. a spec still holding a Custom distribution's closure
    with pytest.raises(ValueError, match="cannot be sent"):
        xd.sample_chains_sharded(spec, 4, dict(num_warmup=10, num_samples=5), devices=[0, 1], engine="oracle_engine")
    with pytest.raises(ValueError):
        xd.sample_chains_sharded(models.eight_schools(), 0, {}, devices=[0])


def test_sharded_api_gives_up_on_ranks_that_hang():
    """A rank that neither finishes nor fails (here: an engine that sleeps in its warmup) must not
    hang the caller: past opts["shard_timeout_s"] the rank processes are terminated and the call raises."""
    import time
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from exmc_amd import distributed as xd
    from exmc_amd import models
    t0 = time.monotonic()
    with pytest.raises(TimeoutError, match="shard_timeout_s"):
        xd.sample_chains_sharded(models.eight_schools(), 4, dict(num_warmup=10, num_samples=5, shard_timeout_s=4.0),
                                 devices=[0, 1], engine="hang_engine")
    assert time.monotonic() - t0 < 60.0


def test_sharded_api_retries_a_failed_peer_on_the_coordinator(monkeypatch):
    """A rank other than the coordinator's fails (an engine that raises in the shard that does not start at chain 0):
    the call runs the chains again on the coordinator's device alone and returns what the ranks would have -- chain i
    keeps its seed -- as the reference retries a failed chain locally (distributed.ex:158-180). With
    retry_on_coordinator off, or when the coordinator's own rank is the one that fails, the failure raises."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from exmc_amd import distributed as xd
    from exmc_amd import models
    monkeypatch.delenv("EXMC_TEST_FAIL_RANK0", raising=False)
    spec = models.eight_schools()
    opts = dict(num_warmup=40, num_samples=12, seed=4, init_values=spec.default_init)
    traces, stats = xd.sample_chains_sharded(spec, 5, opts, devices=[0, 1, 2], engine="flaky_engine")
    t, st = O.sample_chains(O.model_for(spec), 5, init_q=spec.to_unconstrained(spec.default_init),
                            num_warmup=40, num_samples=12, seed=4, cfg=O.Cfg(1, 1))
    extra = stats[0]["extra"]
    assert np.array_equal(extra["raw"]["draws"], t["draws"]) and np.array_equal(extra["raw"]["n_steps"], t["n_steps"])
    assert stats[0]["step_size"] == st.step_size and extra["total_leapfrogs"] == st.total_leapfrogs
    assert extra["retried_on_coordinator"]["rank"] in (1, 2) and "injected failure" in extra["retried_on_coordinator"]["error"]
    assert extra["shards"] == [(0, 5)] and len(traces) == 5
    with pytest.raises(Exception, match="injected failure"):
        xd.sample_chains_sharded(spec, 5, dict(opts, retry_on_coordinator=False), devices=[0, 1, 2], engine="flaky_engine")
    monkeypatch.setenv("EXMC_TEST_FAIL_RANK0", "1")        # (inherited by the spawned ranks)
    with pytest.raises(Exception, match="injected failure"):
        xd.sample_chains_sharded(spec, 5, opts, devices=[0, 1], engine="flaky_engine")
