defmodule Exmc.NUTS.HipNative do
  @moduledoc """
  NIF bindings of `libexmc_hip.so` (the MI355X NUTS inner loop) -- the chain-batched seams the
  Rustler crate `native/exmc_tree` never had. The C side is `c_src/exmc_hip_nif.c`; its
  `ErlNifFunc` table and this module's stubs are compared name by name and arity by arity in
  `tests/test_elixir_sources.py`.

  Conventions are the Rust NIF's (`native/exmc_tree/src/lib.rs:19-32`): binaries are
  native-endian f64, row-major `[chain][step][dim]`; the callee copies in and returns fresh
  binaries; the model handle is a GC'd resource; a decode failure is a `badarg`, a failed
  library call raises `{:exmc_hip_error, code, message}`; every call that waits for the device is
  a dirty (IO-bound) job. Kernel order of the free variables and the flat (PointMap) order are
  related by `model_set_flat_order/2` (`point_map.ex:30-60`).

  Load: `priv/exmc_hip_nif.so` (build line in `INTEGRATION.md`); `EXMC_HIP_DEVICE` selects the GPU.
  """

  @on_load :load_nif

  @doc false
  def load_nif do
    path = :filename.join(:code.priv_dir(:exmc), ~c"exmc_hip_nif")

    case :erlang.load_nif(path, 0) do
      :ok -> :ok
      # the library is optional: without it every stub raises :nif_not_loaded and
      # Exmc.NUTS.HipSampler.available?/0 says false
      {:error, _reason} -> :ok
    end
  end

  @doc "kind: 1 simple, 2 eight_schools, 3 sv, 4 logistic, 5 radon (include/exmc_hip.h); data: f64 binary. -> {:ok, ref} | {:error, message}"
  def model_create(_kind, _data), do: :erlang.nif_error(:nif_not_loaded)

  @doc "A model generated from its Builder IR: path of the plug-in library, the generator's data vector. -> {:ok, ref} | {:error, message}"
  def model_create_plugin(_path, _data), do: :erlang.nif_error(:nif_not_loaded)

  @doc "perm[r] = kernel dimension of flat entry r. -> :ok"
  def model_set_flat_order(_ref, _perm), do: :erlang.nif_error(:nif_not_loaded)

  @doc "vag_fn batched (compiler.ex:131-141): q [C][d] -> {logp [C], grad [C][d]}"
  def logp_grad(_ref, _q, _n_chains), do: :erlang.nif_error(:nif_not_loaded)

  @doc "multi_step_fn (batched_leapfrog.ex:21-48), chain-batched -> {all_q, all_p, all_logp, all_grad}"
  def multi_step(_ref, _q, _p, _grad, _eps, _inv_mass, _n_steps, _n_chains),
    do: :erlang.nif_error(:nif_not_loaded)

  @doc """
  The fused-chain hook of the speculative path (`tree.ex:613-653`): K leapfrog steps of a chain of d <= 256
  independent Normal(mu, sigma) coordinates in one launch. Name, argument order and result of
  `Nx.Vulkan.leapfrog_chain_normal/7`, on f64 binaries `[d]` -> `{:ok, {q_chain, p_chain, grad_chain, logp_chain}}`
  (`[k][d]`, `[k]`); `elixir/patches/tree.ex.diff` adds the `do_dispatch` clause that calls it.
  """
  def leapfrog_chain_normal(_q, _p, _inv_mass, _k, _signed_eps, _mu, _sigma),
    do: :erlang.nif_error(:nif_not_loaded)

  @doc "Shared warmup on chain 0 (sampler.ex:1053-1080) -> %{epsilon, inv_mass, warmup_divergences}"
  def warmup(_ref, _init_q, _num_warmup, _max_tree_depth, _target_accept, _seed),
    do: :erlang.nif_error(:nif_not_loaded)

  @doc "opts[:warm_start] (sampler.ex:167-197): previous step size and inverse mass, min(num_warmup, 50) iterations"
  def warmup_from(_ref, _init_q, _num_warmup, _max_tree_depth, _target_accept, _seed, _prev_epsilon, _prev_inv_mass),
    do: :erlang.nif_error(:nif_not_loaded)

  @doc "opts[:dense_mass] warmup -> %{epsilon, inv_mass, warmup_divergences, cov, chol_cov}; lanes 0 = the kind's layout"
  def warmup_dense(_ref, _init_q, _num_warmup, _max_tree_depth, _target_accept, _seed, _lanes_per_chain),
    do: :erlang.nif_error(:nif_not_loaded)

  @doc "Install a dense mass (flat-order d x d cov and its Cholesky factor) from an earlier run -> :ok"
  def set_dense_mass(_ref, _cov, _chol_cov), do: :erlang.nif_error(:nif_not_loaded)

  @doc "Back to the diagonal mass -> :ok"
  def clear_dense_mass(_ref), do: :erlang.nif_error(:nif_not_loaded)

  @doc """
  Sampling phase of sample_chains_vectorized_compiled (sampler.ex:1082-1130) for chains
  [chain_lo, chain_hi) of n_chains, chain i seeded seed + 7919 i -> {trace_map, leapfrogs, divergences}
  """
  def sample_chains(_ref, _epsilon, _inv_mass, _init_q, _n_chains, _chain_lo, _chain_hi, _num_samples, _max_tree_depth, _seed),
    do: :erlang.nif_error(:nif_not_loaded)

  @doc """
  sample_chains(ir, n, vectorized: false) -- sample_chains_parallel (sampler.ex:1139-1176): every
  chain of [chain_lo, chain_hi) runs its own adaptation and then its draws, all in one launch
  -> {trace_map, tuning_bin, leapfrogs, divergences}; tuning_bin holds 3 + d doubles per chain:
  step size, warmup divergences, warmup leapfrogs, inv_mass (kernel order).
  """
  def sample_independent(_ref, _init_q, _n_chains, _chain_lo, _chain_hi, _num_warmup, _num_samples, _max_tree_depth, _target_accept, _seed),
    do: :erlang.nif_error(:nif_not_loaded)

  @doc "Sampler.sample/3 for one chain (sampler.ex:126-257) -> {trace_map, tuning_map, divergences}"
  def sample(_ref, _init_q, _num_warmup, _num_samples, _max_tree_depth, _target_accept, _seed),
    do: :erlang.nif_error(:nif_not_loaded)

  @doc """
  Sampler.sample/3 with opts[:warm_start] (sampler.ex:167-197): the previous run's step size and inverse
  mass (kernel order), min(num_warmup, 50) warmup iterations, then the chain's draws
  -> {trace_map, tuning_map, divergences}
  """
  def sample_warm(_ref, _init_q, _num_warmup, _num_samples, _max_tree_depth, _target_accept, _seed, _prev_epsilon, _prev_inv_mass),
    do: :erlang.nif_error(:nif_not_loaded)

  @doc """
  Sampler.sample/3 with dense_mass: true (sampler.ex:156, 412-431) -> {trace_map, tuning_map, divergences};
  the tuning map also carries :cov and :chol_cov (d x d row-major f64 binaries, flat order). lanes 0 = the kind's layout.
  """
  def sample_dense(_ref, _init_q, _num_warmup, _num_samples, _max_tree_depth, _target_accept, _seed, _lanes_per_chain),
    do: :erlang.nif_error(:nif_not_loaded)

  @doc "sample_stream/4, pull style: warmup, the chain stays resident -> tuning_map"
  def stream_begin(_ref, _init_q, _num_warmup, _max_tree_depth, _target_accept, _seed),
    do: :erlang.nif_error(:nif_not_loaded)

  @doc "The next n draws of the resident chain -> {trace_map, divergences}"
  def stream_next(_ref, _n_draws), do: :erlang.nif_error(:nif_not_loaded)

  @doc """
  sample_stream/4, push style: ONE launch; a thread of the library sends
  {:exmc_sample, i, q_bin, {tree_depth, n_steps, divergent, accept_prob, energy}} per finished draw
  and {:exmc_done, n, divergences} at the end to pid -> :ok
  """
  def stream_run(_ref, _n_draws, _pid), do: :erlang.nif_error(:nif_not_loaded)
end
