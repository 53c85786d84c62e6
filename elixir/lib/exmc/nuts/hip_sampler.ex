defmodule Exmc.NUTS.HipSampler do
  @moduledoc """
  The HIP branch of `Exmc.NUTS.Sampler` / `Exmc.Compiler.compile_for_sampling/2`: what the two
  patches under `elixir/patches/` call when `Application.get_env(:exmc, :hip, false)`.

  A compiled model is `%{ref: handle, pm: %Exmc.PointMap{}, ncp_info: map, perm: [integer]}`:
  `ref` is a `HipNative` model handle (one of the built kinds, or a plug-in generated from the IR),
  `perm` the kernel dimension of every flat (PointMap) entry. Traces come back from the device as
  `[chain][draw][dim]` f64 binaries in KERNEL order; `flat_draws/4` reorders a chain's rows into the
  flat order `build_trace/3` (`sampler.ex:1281-1298`) slices, so the constrained traces, the
  NCP reconstruction and the stats maps are produced by the reference's own unchanged code.
  """

  alias Exmc.NUTS.{HipExport, HipNative}

  @default_opts [num_warmup: 1000, num_samples: 1000, max_tree_depth: 10, target_accept: 0.8, seed: 0]

  @doc "The probe `Tree.nif_available?/0` (tree.ex:52-54) has, for this library."
  def available? do
    Application.get_env(:exmc, :hip, false) and Code.ensure_loaded?(HipNative) and
      function_exported?(HipNative, :model_create, 2) and nif_loaded?()
  end

  defp nif_loaded? do
    try do
      HipNative.model_create(-1, <<>>)
      true
    rescue
      ErlangError -> false
      ArgumentError -> true
    end
  end

  @doc """
  compile_for_sampling/2, HIP branch (compiler.ex:46-58): export the rewritten IR, generate and build
  the plug-in (cached by the digest of the generated text), load it through the NIF.
  `dir` is a scratch directory for model.json.in / model.json / the library.
  """
  def compile(%Exmc.IR{} = ir, pm, ncp_info, dir, opts \\ []) do
    json = HipExport.to_json(Exmc.Rewrite.apply(ir, ncp: false), Keyword.get(opts, :ncp, true))
    File.mkdir_p!(dir)
    File.write!(Path.join(dir, "model.json.in"), json)
    {_, 0} = System.cmd("python3", ["-m", "exmc_amd.codegen", Path.join(dir, "model.json.in"), dir])
    meta = dir |> Path.join("model.json") |> File.read!() |> Jason.decode!()
    data = for x <- meta["data"], into: <<>>, do: <<x::float-64-native>>
    {:ok, ref} = HipNative.model_create_plugin(Path.join(dir, meta["library"]), data)
    # the generator lays the free variables out in PointMap order already: identity permutation
    %{ref: ref, pm: pm, ncp_info: ncp_info, perm: Enum.to_list(0..(pm.size - 1)), meta: meta}
  end

  @doc "One of the built kinds (include/exmc_hip.h model kinds); `perm` as model_set_flat_order/2 takes it."
  def compile_kind(kind, data_bin, pm, ncp_info, perm) do
    {:ok, ref} = HipNative.model_create(kind, data_bin)
    :ok = HipNative.model_set_flat_order(ref, perm)
    %{ref: ref, pm: pm, ncp_info: ncp_info, perm: perm}
  end

  @doc """
  sample_from_compiled/3 (sampler.ex:126-257), one chain: adaptation and draws as ONE NIF call. opts[:dense_mass]
  (sampler.ex:156) and opts[:warm_start] (sampler.ex:167-197; the `stats` of a previous run: `inv_mass_diag` in
  flat order as every stats map reports it, `step_size`) pick the call. Returns `{tuning, flat draws, trace_map}`;
  `tuning.inv_mass_diag` is the flat-order tensor for the stats map, `tuning.warmup_divergences` the warmup's share
  of `stats.divergences` (sampler.ex:245 counts both phases); a dense tuning also carries `:cov` / `:chol_cov`.
  """
  def sample(%{ref: ref} = compiled, opts) do
    opts = Keyword.merge(@default_opts, opts)
    init_q = init_q(compiled, Keyword.get(opts, :init_q))
    {nw, ns, depth, accept, seed} =
      {opts[:num_warmup], opts[:num_samples], opts[:max_tree_depth], opts[:target_accept], opts[:seed]}

    {trace, tuning, _divergences} =
      cond do
        Keyword.get(opts, :dense_mass, false) ->
          HipNative.sample_dense(ref, init_q, nw, ns, depth, accept, seed, 0)

        ws = Keyword.get(opts, :warm_start) ->
          prev_im = kernel_inv_mass(compiled, ws.inv_mass_diag)
          HipNative.sample_warm(ref, init_q, nw, ns, depth, accept, seed, ws.step_size * 1.0, prev_im)

        true ->
          HipNative.sample(ref, init_q, nw, ns, depth, accept, seed)
      end

    tuning = Map.put(tuning, :inv_mass_diag, flat_inv_mass(compiled, tuning.inv_mass))
    [draws] = chains(compiled, trace, 1, opts[:num_samples])
    {tuning, draws, trace}
  end

  @doc """
  sample_from_compiled_tuned/4 (sampler.ex:259-335; `sample_compiled_tuned/4`, and what
  `Distributed.run_chain_remote` calls): no warmup, the draws of one chain under the tuning map
  `%{epsilon, inv_mass, chol_cov}` of the reference -- `inv_mass` a flat-order `{d}` tensor, or the `{d, d}`
  covariance together with its Cholesky factor `chol_cov` for a dense mass (sampler.ex:274, 292).
  Returns `{flat draws, trace_map}`.
  """
  def sample_tuned(%{ref: ref, pm: pm} = compiled, tuning, opts) do
    opts = Keyword.merge(@default_opts, opts)
    init_q = init_q(compiled, Keyword.get(opts, :init_q))
    d = pm.size

    diag =
      case Map.get(tuning, :chol_cov) do
        nil ->
          :ok = HipNative.clear_dense_mass(ref)
          tuning.inv_mass

        chol ->
          f64 = fn t -> t |> Nx.as_type(:f64) |> Nx.reshape({d * d}) |> Nx.to_binary() end
          :ok = HipNative.set_dense_mass(ref, f64.(tuning.inv_mass), f64.(chol))
          Nx.take_diagonal(tuning.inv_mass)
      end

    {trace, _leapfrogs, _divergences} =
      HipNative.sample_chains(ref, tuning.epsilon * 1.0, kernel_inv_mass(compiled, diag), init_q, 1, 0, 1,
        opts[:num_samples], opts[:max_tree_depth], opts[:seed])

    [draws] = chains(compiled, trace, 1, opts[:num_samples])
    {draws, trace}
  end

  @doc """
  sample_chains_vectorized_compiled/3 (sampler.ex:1020-1136): the shared warmup on chain 0, then
  every chain with its tuning, as two NIF calls. Returns `{tuning, [flat draws per chain], trace_map}`;
  the caller builds traces and stats with its own build_trace/3.
  """
  def sample_chains_vectorized(%{ref: ref} = compiled, num_chains, opts) do
    opts = Keyword.merge(@default_opts, opts)
    init_q = init_q(compiled, Keyword.get(opts, :init_q))

    tuning =
      HipNative.warmup(ref, init_q, opts[:num_warmup], opts[:max_tree_depth], opts[:target_accept], opts[:seed])

    {trace, _leapfrogs, _divergences} =
      HipNative.sample_chains(ref, tuning.epsilon, tuning.inv_mass, init_q, num_chains, 0, num_chains,
        opts[:num_samples], opts[:max_tree_depth], opts[:seed])

    # tuning.inv_mass stays in KERNEL order (it goes back into HipNative.sample_chains above);
    # stats.inv_mass_diag is reported in the flat order of the trace, as the reference's is
    tuning = Map.put(tuning, :inv_mass_diag, flat_inv_mass(compiled, tuning.inv_mass))
    {tuning, chains(compiled, trace, num_chains, opts[:num_samples]), trace}
  end

  @doc """
  sample_chains_parallel (sampler.ex:1139-1176), `vectorized: false`: every chain its own adaptation,
  one launch. Returns `{[tuning per chain], [flat draws per chain], trace_map}`.
  """
  def sample_chains_independent(%{ref: ref, pm: pm} = compiled, num_chains, opts) do
    opts = Keyword.merge(@default_opts, opts)
    init_q = init_q(compiled, Keyword.get(opts, :init_q))

    {trace, tuning_bin, _leapfrogs, _divergences} =
      HipNative.sample_independent(ref, init_q, num_chains, 0, num_chains, opts[:num_warmup], opts[:num_samples],
        opts[:max_tree_depth], opts[:target_accept], opts[:seed])

    d = pm.size
    row = (3 + d) * 8

    tunings =
      for c <- 0..(num_chains - 1) do
        <<eps::float-64-native, wdiv::float-64-native, _wlf::float-64-native, im::binary-size(d * 8)>> =
          binary_part(tuning_bin, c * row, row)

        %{epsilon: eps, warmup_divergences: trunc(wdiv), inv_mass: im, inv_mass_diag: flat_inv_mass(compiled, im)}
      end

    {tunings, chains(compiled, trace, num_chains, opts[:num_samples]), trace}
  end

  @doc "sample_stream/4 (sampler.ex:1186-1277): warmup, then ONE launch whose draws arrive as messages."
  def sample_stream(%{ref: ref} = compiled, receiver_pid, opts) do
    opts = Keyword.merge(@default_opts, opts)
    init_q = init_q(compiled, Keyword.get(opts, :init_q))
    _tuning = HipNative.stream_begin(ref, init_q, opts[:num_warmup], opts[:max_tree_depth], opts[:target_accept], opts[:seed])
    HipNative.stream_run(ref, opts[:num_samples], receiver_pid)
  end

  # ---- layout helpers ----

  # [chain][draw][dim] kernel-order binary -> per chain a list of flat-order f64 row tensors
  defp chains(%{pm: pm, perm: perm}, %{draws: draws}, num_chains, num_samples) do
    d = pm.size
    chain_bytes = num_samples * d * 8

    for c <- 0..(num_chains - 1) do
      flat_draws(binary_part(draws, c * chain_bytes, chain_bytes), num_samples, d, perm)
    end
  end

  @doc """
  A kernel-order inverse mass diagonal (f64 binary, as the NIF returns it) as the `{d}` tensor
  `stats.inv_mass_diag` carries: flat (PointMap) order, the order of the trace. For the built kinds
  `perm` is not the identity (e.g. radon's counties by size, sv's string sort).
  """
  def flat_inv_mass(%{perm: perm}, inv_mass_bin) do
    Nx.from_binary(inv_mass_bin, :f64) |> Nx.take(Nx.tensor(perm, type: :s64))
  end

  @doc """
  The inverse of `flat_inv_mass/2`: a flat-order `{d}` tensor (what `stats.inv_mass_diag` holds, e.g. inside
  opts[:warm_start]) as the kernel-order f64 binary the NIFs take. Kernel dimension perm[r] takes flat entry r.
  """
  def kernel_inv_mass(%{perm: perm}, %Nx.Tensor{} = flat) do
    vals = flat |> Nx.as_type(:f64) |> Nx.to_flat_list()

    perm
    |> Enum.zip(vals)
    |> Enum.sort()
    |> Enum.reduce(<<>>, fn {_k, x}, acc -> <<acc::binary, x::float-64-native>> end)
  end

  @doc false
  def flat_draws(chain_bin, num_samples, d, perm) do
    t = Nx.from_binary(chain_bin, :f64) |> Nx.reshape({num_samples, d})
    idx = Nx.tensor(perm, type: :s64)
    # flat entry r = kernel dimension perm[r]
    Nx.take(t, idx, axis: 1)
  end

  # opts[:init_q]: the flat unconstrained start the sampler's own init_position/5 computed from
  # opts[:init_values] (sampler.ex:351-356: NCP inversion, to_unconstrained, pack), or nil for the
  # random start (0.1 * normal_s per flat entry, drawn on the device from the chain's seed)
  defp init_q(_compiled, nil), do: nil

  defp init_q(%{perm: perm}, %Nx.Tensor{} = flat_q) do
    flat = flat_q |> Nx.as_type(:f64) |> Nx.to_flat_list()
    # kernel dimension perm[r] takes flat entry r
    kernel =
      perm
      |> Enum.with_index()
      |> Enum.sort()
      |> Enum.map(fn {_k, r} -> Enum.at(flat, r) end)

    for x <- kernel, into: <<>>, do: <<x::float-64-native>>
  end
end
