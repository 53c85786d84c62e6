"""sampler.sample_chains(ir, n, devices=[...]): the chain-sharding fan-out as an API call
(exmc_amd.distributed.sample_chains_sharded, the analogue of Exmc.NUTS.Distributed.sample_chains/2).
On the one-GPU box both rank processes use device 0; the sharded result equals the one-process
result bit for bit."""
import numpy as np
import pytest

from exmc_amd import models, sampler

pytestmark = pytest.mark.gpu


def test_two_rank_processes_equal_one_process(hip):
    spec = models.eight_schools()
    opts = dict(num_warmup=100, num_samples=40, seed=21, init_values=spec.default_init)
    t1, s1 = sampler.sample_chains(spec, 7, opts)
    t2, s2 = sampler.sample_chains(spec, 7, dict(opts, devices=[0, 0]))
    r1, r2 = s1[0]["extra"]["raw"], s2[0]["extra"]["raw"]
    for k in ("draws", "logp", "tree_depth", "n_steps", "divergent", "accept_prob", "energy"):
        assert np.array_equal(r1[k], r2[k]), k
    assert s1[0]["step_size"] == s2[0]["step_size"]
    assert s2[0]["extra"]["shards"] == [(0, 4), (4, 7)]
    assert s1[0]["extra"]["total_leapfrogs"] == s2[0]["extra"]["total_leapfrogs"]
    assert all(np.array_equal(t1[c]["tau"], t2[c]["tau"]) for c in range(7))


def test_sharded_dense_mass_returns_the_dense_tuning(hip):
    """opts["dense_mass"] through the fan-out: every rank repeats the dense warmup, the covariance and
    its factor come back as the single-device path returns them, inv_mass_diag stays a vector."""
    spec = models.eight_schools()
    opts = dict(num_warmup=150, num_samples=20, seed=5, init_values=spec.default_init, dense_mass=True,
                lanes_per_chain=16)
    t1, s1 = sampler.sample_chains(spec, 4, opts)
    t2, s2 = sampler.sample_chains(spec, 4, dict(opts, devices=[0, 0]))
    assert np.array_equal(s1[0]["extra"]["raw"]["draws"], s2[0]["extra"]["raw"]["draws"])
    assert s2[0]["inv_mass_diag"].shape == (spec.d,) and s2[0]["chol_cov"].shape == (spec.d, spec.d)
    assert np.array_equal(s2[0]["cov"], s1[0]["cov"]) if "cov" in s1[0] else True
