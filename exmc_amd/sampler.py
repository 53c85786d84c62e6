"""Host-side mirror of Exmc.NUTS.Sampler (lib/exmc/nuts/sampler.ex) over libexmc_hip.so.

Same entry points, option names, return shapes and error behaviour as the reference module:
    sample(ir, init_values, opts)            -> {trace, stats}          sampler.ex:33-37
    sample_chains(ir, num_chains, opts)      -> {[trace], [stats]}      sampler.ex:992-1000
    sample_stream(ir, receiver, init, opts)  -> :ok + messages          sampler.ex:1186-1277
    compile / sample_compiled / sample_compiled_tuned / sample_chains_compiled
The "ir" here is an exmc_amd.models.ModelSpec. All arithmetic of the hot path runs in the HIP
kernels; this file only marshals buffers. There is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib
from .models import ModelSpec

DEFAULT_OPTS = dict(num_warmup=1000, num_samples=1000, max_tree_depth=10, target_accept=0.8,
                    seed=0, supervised=False)  # sampler.ex:16-23


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Compiled:
    """Compiler.compile_for_sampling/2 result (compiler.ex:46-58): owns the device handle."""

    def __init__(self, spec, device=0):
        if not isinstance(spec, ModelSpec):
            raise TypeError("expected a ModelSpec")
        self.spec = spec
        # a generated model brings its own build of the library (codegen.GeneratedSpec.lib_path)
        lib_path = getattr(spec, "lib_path", None)
        L = self.L = _lib.bind(lib_path) if lib_path else _lib.load()
        h = C.c_void_p()
        _lib.check(L.exmc_hip_model_create(spec.kind, spec.d, _dp(spec.data), int(spec.data.size),
                                           int(device), C.byref(h)), L)
        self.h = h
        self.d = spec.d
        self.device = device
        # PointMap.build's layout (point_map.ex:30-60): the RNG-consuming steps draw in the flat
        # order of the sorted ids; the kernels keep their own compute layout
        order = np.ascontiguousarray(spec.flat_order(), dtype=np.int32)
        _lib.check(L.exmc_hip_model_set_flat_order(
            h, order.ctypes.data_as(C.POINTER(C.c_int32)), int(order.size)), L)

    def close(self):
        if getattr(self, "h", None):
            self.L.exmc_hip_model_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        _lib.check(rc, self.L)

    @property
    def default_lanes(self):
        return self.L.exmc_hip_model_default_lanes(self.h)

    @property
    def default_warmup_lanes(self):
        return self.L.exmc_hip_model_default_warmup_lanes(self.h)

    @property
    def default_dense_lanes(self):
        return self.L.exmc_hip_model_default_dense_lanes(self.h)

    @property
    def last_kernel_ms(self):
        return self.L.exmc_hip_last_kernel_ms(self.h)


def compile(ir, opts=None):  # noqa: A001  (name mirrors Sampler.compile/2)
    opts = opts or {}
    return Compiled(ir, device=opts.get("device", 0))


def _merge_opts(opts):
    o = dict(DEFAULT_OPTS)
    o.update(opts or {})
    return o


def _c_opts(o, lanes=None):
    return _lib.Opts(int(o["num_warmup"]), int(o["num_samples"]), int(o["max_tree_depth"]),
                     float(o["target_accept"]), int(o["seed"]),
                     int(lanes if lanes is not None else o.get("lanes_per_chain", 0)))


def _init_q(spec, init_values):
    if not init_values:
        return None  # sampler.ex:339-349: 0.1 * normal_s per dimension
    return np.ascontiguousarray(spec.to_unconstrained(init_values))


def _host_trace(n_chains, num_samples, d):
    n = n_chains * num_samples
    t = dict(draws=np.zeros((n_chains, num_samples, d)), logp=np.zeros((n_chains, num_samples)),
             tree_depth=np.zeros((n_chains, num_samples), np.int32),
             n_steps=np.zeros((n_chains, num_samples), np.int32),
             divergent=np.zeros((n_chains, num_samples), np.int32),
             accept_prob=np.zeros((n_chains, num_samples)),
             energy=np.zeros((n_chains, num_samples)))
    assert n >= 0
    tr = _lib.Trace(*[t[k].ctypes.data for k in
                      ("draws", "logp", "tree_depth", "n_steps", "divergent", "accept_prob",
                       "energy")])
    return t, tr


class SampleStats:
    """stats.sample_stats (sampler.ex:960-967): a sequence of per-draw maps, backed by arrays."""

    def __init__(self, raw, chain):
        self._raw = raw
        self._c = chain

    def __len__(self):
        return self._raw["tree_depth"].shape[1]

    def __getitem__(self, i):
        r, c = self._raw, self._c
        return dict(tree_depth=int(r["tree_depth"][c, i]), n_steps=int(r["n_steps"][c, i]),
                    divergent=bool(r["divergent"][c, i]), accept_prob=float(r["accept_prob"][c, i]),
                    energy=float(r["energy"][c, i]), recovered=False)

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def column(self, key):
        return self._raw[key][self._c]


def _build_trace(spec, draws):
    """build_trace (sampler.ex:1281-1298): slice per entry + forward transform."""
    x = spec.constrain(draws)
    simplex = getattr(spec, "simplex_entries", {})
    inside = {off + i for off, n in simplex.values() for i in range(n)}
    trace = {name: x[:, i] for i, name in enumerate(spec.var_names) if i not in inside}
    # a vector rv is one PointMap entry: its draws also come back as one {S, length} array under its id
    for id_, (off, n) in getattr(spec, "vector_entries", {}).items():
        if id_ in simplex:    # K - 1 unconstrained slots -> the K-simplex (Transform.apply(:stick_breaking))
            from .codegen import stick_breaking
            trace[id_] = stick_breaking(x[:, off:off + n])
        else:
            trace[id_] = x[:, off:off + n]
    return trace


def _tuning_struct(tuning, d):
    t = _lib.Tuning()
    t.epsilon = float(tuning["epsilon"])
    im = np.asarray(tuning["inv_mass"], dtype=np.float64)
    if im.ndim == 2 and im.shape == (d, d):
        im = np.diag(im).copy()   # a dense tuning carries cov; the struct takes its diagonal (sampler.ex:236-240)
    if im.ndim != 1 or im.shape[0] != d:
        raise ValueError("inv_mass must be a rank-1 tensor of length d, or d x d with chol_cov for a dense mass")
    for i in range(d):
        t.inv_mass[i] = im[i]
    return t


def sample_compiled(compiled, init_values=None, opts=None):
    """sample_from_compiled (sampler.ex:126-257), diagonal mass. opts["warm_start"] =
    {"inv_mass_diag": ..., "step_size": ...} (the `stats` of a previous run carry both keys)
    reuses that tuning and runs only min(num_warmup, 50) warmup iterations (sampler.ex:167-197)."""
    o = _merge_opts(opts)
    spec = compiled.spec
    L = compiled.L
    t, tr = _host_trace(1, o["num_samples"], spec.d)
    tun = _lib.Tuning()
    div = C.c_int32()
    iq = _init_q(spec, init_values)
    if o.get("dense_mass"):
        d = spec.d
        cov, chol = np.zeros((d, d)), np.zeros((d, d))
        compiled.check(L.exmc_hip_sample_dense_host(compiled.h, None if iq is None else _dp(iq),
                                                    _c_opts(o, lanes=o.get("lanes_per_chain") or compiled.default_dense_lanes),
                                                    tr, C.byref(tun), _dp(cov), _dp(chol), C.byref(div)))
        trace = _build_trace(spec, t["draws"][0])
        stats = dict(step_size=tun.epsilon, inv_mass_diag=np.array(tun.inv_mass[:d]), chol_cov=chol, cov=cov,
                     divergences=int(div.value), recoveries=0, num_warmup=o["num_warmup"],
                     num_samples=o["num_samples"], sample_stats=SampleStats(t, 0), raw=t)
        return trace, stats
    compiled.check(L.exmc_hip_model_clear_dense_mass(compiled.h))
    ws = o.get("warm_start")
    start = None
    if ws is not None:
        start = C.byref(_tuning_struct(dict(epsilon=ws["step_size"], inv_mass=ws["inv_mass_diag"]), spec.d))
    compiled.check(L.exmc_hip_sample_warm_host(compiled.h, None if iq is None else _dp(iq), _c_opts(o),
                                               start, tr, C.byref(tun), C.byref(div)))
    trace = _build_trace(spec, t["draws"][0])
    stats = dict(step_size=tun.epsilon, inv_mass_diag=np.array(tun.inv_mass[:spec.d]),
                 divergences=int(div.value), recoveries=0, num_warmup=o["num_warmup"],
                 num_samples=o["num_samples"], sample_stats=SampleStats(t, 0), raw=t)
    return trace, stats


def sample(ir, init_values=None, opts=None):
    """Exmc.NUTS.Sampler.sample/3."""
    opts = opts or {}
    compiled = ir if isinstance(ir, Compiled) else Compiled(ir, device=opts.get("device", 0))
    return sample_compiled(compiled, init_values, opts)


def _apply_mass(compiled, tuning):
    """Install the mass matrix a tuning map names: dense when it carries :chol_cov (sampler.ex:274,
    292), diagonal otherwise."""
    L, d = compiled.L, compiled.d
    chol = tuning.get("chol_cov")
    if chol is None:
        compiled.check(L.exmc_hip_model_clear_dense_mass(compiled.h))
        return
    cov = np.ascontiguousarray(tuning.get("cov", tuning["inv_mass"]), dtype=np.float64)
    chol = np.ascontiguousarray(chol, dtype=np.float64)
    if cov.shape != (d, d) or chol.shape != (d, d):
        raise ValueError("a dense tuning needs cov and chol_cov of shape (d, d)")
    compiled.check(L.exmc_hip_model_set_dense_mass(compiled.h, _dp(cov), _dp(chol), d))


def warmup(compiled, init_values=None, opts=None):
    """Shared warmup on chain 0 (sampler.ex:1053-1080); returns the tuning map. With
    opts["dense_mass"] the map carries the covariance under "cov" (and as "inv_mass", the d x d
    M^-1 the reference's tuning holds) and its Cholesky factor under "chol_cov" (sampler.ex:65)."""
    o = _merge_opts(opts)
    L = compiled.L
    tun = _lib.Tuning()
    iq = _init_q(compiled.spec, init_values)
    if o.get("dense_mass"):
        d = compiled.d
        cov, chol = np.zeros((d, d)), np.zeros((d, d))
        compiled.check(L.exmc_hip_warmup_dense(compiled.h, None if iq is None else _dp(iq),
                                               _c_opts(o, lanes=o.get("lanes_per_chain") or compiled.default_dense_lanes),
                                               C.byref(tun), _dp(cov), _dp(chol)))
        return dict(epsilon=tun.epsilon, inv_mass=cov, cov=cov, chol_cov=chol,
                    inv_mass_diag=np.array(tun.inv_mass[:d]), warmup_divergences=tun.warmup_divergences)
    compiled.check(L.exmc_hip_model_clear_dense_mass(compiled.h))
    # the layout of the one-chain warmup: opts["warmup_lanes"], else the sampling layout when the
    # caller named one, else the library's choice for this model (its tuning is layout-independent)
    lanes = o.get("warmup_lanes") or o.get("lanes_per_chain") or compiled.default_warmup_lanes
    compiled.check(L.exmc_hip_warmup(compiled.h, None if iq is None else _dp(iq), _c_opts(o, lanes=lanes),
                                 C.byref(tun)))
    return dict(epsilon=tun.epsilon, inv_mass=np.array(tun.inv_mass[:compiled.d]), chol_cov=None,
                warmup_divergences=tun.warmup_divergences)


def sample_compiled_tuned(compiled, tuning, init_values=None, opts=None, num_chains=1,
                          chain_lo=0, chain_hi=None):
    """sample_compiled_tuned/4 (sampler.ex:69-71, 260-335), generalised to a chain range: chain i
    uses seed + 7919*i (sampler.ex:1083)."""
    o = _merge_opts(opts)
    spec = compiled.spec
    L = compiled.L
    chain_hi = num_chains if chain_hi is None else chain_hi
    nc = chain_hi - chain_lo
    tun = _tuning_struct(tuning, spec.d)
    _apply_mass(compiled, tuning)
    if tuning.get("chol_cov") is not None and not o.get("lanes_per_chain"):
        o["lanes_per_chain"] = compiled.default_dense_lanes   # the layout this kind carries a dense mass in
    lf = C.c_int64()
    dv = C.c_int32()
    iq = _init_q(spec, init_values)
    if o.get("raw_on_device"):
        # the finished trace stays where the kernel wrote it: torch tensors in the device layout
        # [S][d][C] / [S][C] under extra["raw"] (exmc_hip_sample_chains, caller-owned device buffers);
        # the sharded fan-out gathers them over RCCL from there (exmc_amd/distributed.py). No
        # per-chain host traces are built on this path.
        import torch
        dev = torch.device("cuda", compiled.device)
        S = int(o["num_samples"])
        f64, i32 = torch.float64, torch.int32
        raw = dict(draws=torch.empty((S, spec.d, nc), dtype=f64, device=dev), logp=torch.empty((S, nc), dtype=f64, device=dev),
                   tree_depth=torch.empty((S, nc), dtype=i32, device=dev), n_steps=torch.empty((S, nc), dtype=i32, device=dev),
                   divergent=torch.empty((S, nc), dtype=i32, device=dev), accept_prob=torch.empty((S, nc), dtype=f64, device=dev),
                   energy=torch.empty((S, nc), dtype=f64, device=dev))
        trd = _lib.Trace(*[raw[k].data_ptr() for k in ("draws", "logp", "tree_depth", "n_steps", "divergent",
                                                       "accept_prob", "energy")])
        compiled.check(L.exmc_hip_sample_chains(compiled.h, C.byref(tun), None if iq is None else _dp(iq),
                                                num_chains, chain_lo, chain_hi, _c_opts(o), trd, C.byref(lf),
                                                C.byref(dv)))
        return None, None, dict(total_leapfrogs=int(lf.value), total_divergences=int(dv.value), raw=raw,
                                kernel_ms=compiled.last_kernel_ms)
    t, tr = _host_trace(nc, o["num_samples"], spec.d)
    compiled.check(L.exmc_hip_sample_chains_host(compiled.h, C.byref(tun),
                                             None if iq is None else _dp(iq), num_chains,
                                             chain_lo, chain_hi, _c_opts(o), tr, C.byref(lf),
                                             C.byref(dv)))
    traces, stats = [], []
    for c in range(nc):
        traces.append(_build_trace(spec, t["draws"][c]))
        stats.append(dict(step_size=tun.epsilon, inv_mass_diag=np.array(tun.inv_mass[:spec.d]),
                          divergences=int(t["divergent"][c].sum()), num_warmup=o["num_warmup"],
                          num_samples=o["num_samples"], sample_stats=SampleStats(t, c)))
    extra = dict(total_leapfrogs=int(lf.value), total_divergences=int(dv.value), raw=t,
                 kernel_ms=compiled.last_kernel_ms)
    return traces, stats, extra


def sample_chains_compiled(compiled, num_chains, opts=None):
    """sample_chains_vectorized_compiled (sampler.ex:1020-1136): warmup once on chain 0, then all
    chains share epsilon and the mass matrix. On the GPU the chains run as one batch."""
    if num_chains < 1:
        raise ValueError("num_chains must be >= 1")
    o = _merge_opts(opts)
    init_values = o.get("init_values") or {}
    tuning = warmup(compiled, init_values, o)
    traces, stats, extra = sample_compiled_tuned(compiled, tuning, init_values, o,
                                                 num_chains=num_chains)
    for s in stats:
        s["extra"] = extra
    return traces, stats


def sample_chains_independent_compiled(compiled, num_chains, opts=None, chain_lo=0, chain_hi=None):
    """sample_chains_parallel (sampler.ex:1139-1176) -- `sample_chains(ir, n, vectorized: false)`:
    chain i is Sampler.sample/3 with seed + 7919 i, its own adaptation and then its draws. On the
    GPU all chains of [chain_lo, chain_hi) run as ONE launch, every lane group a chain from its first
    warmup transition to its last draw (exmc_hip_sample_independent_host). Each chain's stats carry
    its own step size, inverse mass and warmup divergences, as the reference's per-chain stats do."""
    if num_chains < 1:
        raise ValueError("num_chains must be >= 1")
    o = _merge_opts(opts)
    chain_hi = num_chains if chain_hi is None else chain_hi
    nc = chain_hi - chain_lo
    if o.get("dense_mass") or o.get("warm_start"):
        # The one-launch kernel adapts a diagonal mass from a cold start. The reference forwards
        # every sample/3 option to each chain (sampler.ex:1146-1153), so these two run as what they
        # are there: sample_from_compiled per chain, seed + 7919 i, one after the other.
        drop = ("init_values", "parallel", "max_concurrency", "vectorized", "devices")
        base = {k: v for k, v in o.items() if k not in drop}
        init_values = o.get("init_values") or {}
        traces, stats = [], []
        for i in range(chain_lo, chain_hi):
            tr, st = sample_compiled(compiled, init_values, dict(base, seed=int(o["seed"]) + 7919 * i))
            traces.append(tr)
            stats.append(st)
        return traces, stats
    spec = compiled.spec
    L = compiled.L
    iq = _init_q(spec, o.get("init_values") or {})
    lf, dv = C.c_int64(), C.c_int32()
    tune = np.zeros((nc, 3 + spec.d))
    t, tr = _host_trace(nc, o["num_samples"], spec.d)
    compiled.check(L.exmc_hip_sample_independent_host(compiled.h, None if iq is None else _dp(iq), num_chains,
                                                      chain_lo, chain_hi, _c_opts(o), tr, _dp(tune),
                                                      C.byref(lf), C.byref(dv)))
    traces, stats = [], []
    extra = dict(total_leapfrogs=int(lf.value), total_divergences=int(dv.value), raw=t,
                 kernel_ms=compiled.last_kernel_ms, tuning=tune,
                 warmup_leapfrogs=int(tune[:, 2].sum()))
    for c in range(nc):
        traces.append(_build_trace(spec, t["draws"][c]))
        # stats.divergences counts warmup + sampling (sampler.ex:245)
        stats.append(dict(step_size=float(tune[c, 0]), inv_mass_diag=np.array(tune[c, 3:]),
                          divergences=int(t["divergent"][c].sum()) + int(tune[c, 1]),
                          num_warmup=o["num_warmup"], num_samples=o["num_samples"],
                          sample_stats=SampleStats(t, c), extra=extra))
    return traces, stats


def sample_chains(ir, num_chains, opts=None):
    """Exmc.NUTS.Sampler.sample_chains/3. opts["vectorized"] (default: num_chains > 1, sampler.ex:993)
    chooses between the shared warmup (one adaptation, every chain with its tuning) and
    sample_chains_parallel (False: every chain adapts on its own, sample_chains_independent_compiled).
    With opts["devices"] = [0, 1, ...] the chains are sharded over those GPUs, one process each
    (exmc_amd.distributed.sample_chains_sharded, the analogue of Exmc.NUTS.Distributed.sample_chains/2);
    the result does not depend on the number of devices. A peer rank that fails has its call's chains run
    again on the coordinator's device (devices[0]), as the reference retries a failed chain on the
    coordinator (distributed.ex:158-180; opts["retry_on_coordinator"] = False: raise instead); a failing
    rank 0 and ranks that hang past opts["shard_timeout_s"] take the call down."""
    opts = opts or {}
    devices = opts.get("devices")
    vectorized = opts.get("vectorized", num_chains > 1)
    if not (vectorized and num_chains > 1):
        if devices is not None and len(devices) > 1 and num_chains > 1:
            raise ValueError("vectorized: false is a single-device mode here (shard the chain range yourself: "
                             "sample_chains_independent_compiled(..., chain_lo, chain_hi))")
        # (one chain has nothing to shard: it runs on the first device named)
        o1 = dict(opts, device=devices[0]) if devices else opts
        compiled = ir if isinstance(ir, Compiled) else Compiled(ir, device=o1.get("device", 0))
        return sample_chains_independent_compiled(compiled, num_chains, o1)
    if devices is not None and len(devices) > 1:
        if isinstance(ir, Compiled):
            ir = ir.spec
        from . import distributed
        return distributed.sample_chains_sharded(ir, num_chains, opts, devices)
    if devices:
        opts = dict(opts, device=devices[0])
    compiled = ir if isinstance(ir, Compiled) else Compiled(ir, device=opts.get("device", 0))
    return sample_chains_compiled(compiled, num_chains, opts)


def sample_stream(ir, receiver, init_values=None, opts=None):
    """Exmc.NUTS.Sampler.sample_stream/4: `receiver` is called with the reference's messages,
    ("exmc_sample", i, point_map, step_stat) for i = 1..n then ("exmc_done", n). The warmup runs
    first; the draws are then produced `stream_chunk` (default 50) at a time by the resident chain
    (exmc_hip_stream_begin / _next_host) and delivered as each chunk lands, so the receiver sees
    samples while the chain is still running. opts["stream_push"]: one launch for all draws that
    writes every finished draw into page-locked host memory and publishes its count
    (exmc_hip_stream_start / _finish); this function polls the count and delivers each draw as it
    appears -- the per-draw notification of sampler.ex:1240-1270 without a launch per message.
    Either way the stream equals sample/3's draws bit for bit."""
    o = _merge_opts(opts)
    compiled = ir if isinstance(ir, Compiled) else Compiled(ir, device=o.get("device", 0))
    spec = compiled.spec
    L = compiled.L
    n = int(o["num_samples"])
    chunk = max(1, int(o.get("stream_chunk", 50)))
    tun = _lib.Tuning()
    iq = _init_q(spec, init_values)
    # (stream_begin runs the diagonal adaptation and clears a dense mass an earlier call left on the
    # handle: the stream equals sample/3 whatever ran before)
    compiled.check(L.exmc_hip_stream_begin(compiled.h, None if iq is None else _dp(iq), _c_opts(o),
                                       C.byref(tun)))
    sent = 0
    if o.get("stream_push"):
        import time
        view = _lib.Trace()
        prog = C.POINTER(C.c_int32)()
        compiled.check(L.exmc_hip_stream_start(compiled.h, n, C.byref(view), C.byref(prog)))

        def col(ptr, ctype, width=1):
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n * width,))
            return a.reshape(n, width) if width > 1 else a
        draws = col(view.draws, C.c_double, spec.d)
        raw = dict(tree_depth=col(view.tree_depth, C.c_int32)[None, :], n_steps=col(view.n_steps, C.c_int32)[None, :],
                   divergent=col(view.divergent, C.c_int32)[None, :], accept_prob=col(view.accept_prob, C.c_double)[None, :],
                   energy=col(view.energy, C.c_double)[None, :])
        seen = []                                  # the counts this poll loop observed (tests look at it)
        deadline = time.monotonic() + float(o.get("stream_timeout_s", 600.0))
        # Whatever happens below (a timeout, the receiver raising), the run is finished on the
        # library side before this function returns: the launch drains, the handle leaves its
        # "stream run in flight" state, and nothing handed to the receiver aliases the page-locked
        # trace the next run may reuse or free (rows are copied out first).
        try:
            while sent < n:
                ready = int(prog[0])               # rows [0, ready) are final (system-scope release on the device)
                if ready > sent:
                    seen.append(ready)
                    x = spec.constrain(np.array(draws[sent:ready]))
                    rows = SampleStats({k: np.array(v[:, sent:ready]) for k, v in raw.items()}, 0)
                    for i in range(sent, ready):
                        point_map = {name: float(x[i - sent, j]) for j, name in enumerate(spec.var_names)}
                        receiver(("exmc_sample", i + 1, point_map, rows[i - sent]))
                    sent = ready
                elif time.monotonic() > deadline:
                    raise TimeoutError("stream: no draw within stream_timeout_s")
                else:
                    time.sleep(0.0002)
        finally:
            div = C.c_int32()
            rc = L.exmc_hip_stream_finish(compiled.h, C.byref(div))
        compiled.check(rc)
        receiver(("exmc_done", n))
        compiled.last_stream_counts = seen
        return "ok"
    while sent < n:
        m = min(chunk, n - sent)
        t, tr = _host_trace(1, m, spec.d)
        div = C.c_int32()
        compiled.check(L.exmc_hip_stream_next_host(compiled.h, m, tr, C.byref(div)))
        x = spec.constrain(t["draws"][0])
        ss = SampleStats(t, 0)
        for i in range(m):
            point_map = {name: float(x[i, j]) for j, name in enumerate(spec.var_names)}
            receiver(("exmc_sample", sent + i + 1, point_map, ss[i]))
        sent += m
    receiver(("exmc_done", n))
    return "ok"
