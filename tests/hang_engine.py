"""An engine whose ranks never finish (test infrastructure): sample_chains_sharded must end them and
raise instead of polling forever."""
import time

from exmc_amd.sampler import SampleStats, _build_trace, _merge_opts  # noqa: F401


def compile(spec, opts=None):  # noqa: A001
    return object()


def warmup(compiled, init_values=None, opts=None):
    time.sleep(3600)
