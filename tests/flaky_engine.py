"""An engine (the CPU checker's, tests/oracle_engine.py) whose rank fails when its shard does not start at chain 0
(EXMC_TEST_FAIL_RANK0 unset) or does (set): what sample_chains_sharded does about a failed peer and a failed
coordinator (the reference: lib/exmc/nuts/distributed.ex:158-180). Test infrastructure only."""
import os

from oracle_engine import SampleStats, _build_trace, _merge_opts, compile, warmup  # noqa: F401
from oracle_engine import sample_compiled_tuned as _sample


def sample_compiled_tuned(compiled, tuning, init_values=None, opts=None, num_chains=1, chain_lo=0, chain_hi=None):
    whole = chain_lo == 0 and (chain_hi is None or chain_hi == num_chains)
    fail_rank0 = bool(os.environ.get("EXMC_TEST_FAIL_RANK0"))
    if not whole and ((chain_lo == 0) == fail_rank0):
        raise RuntimeError("injected failure of the rank that owns chains [%d, %s)" % (chain_lo, chain_hi))
    return _sample(compiled, tuning, init_values, opts, num_chains=num_chains, chain_lo=chain_lo, chain_hi=chain_hi)
