defmodule Exmc.NUTS.HipExport do
  @moduledoc """
  `%Exmc.IR{}` -> JSON for `python -m exmc_amd.codegen` (the generator restates the term walk of
  `lib/exmc/compiler.ex:176-269` and writes HIP text; `INTEGRATION.md` section 2a).

  The IR is exported after `Exmc.Rewrite.apply/2` (`rewrite.ex:23-34`: transforms explicit). Nodes
  are `%Exmc.Node{id, op}` with `op` one of `{:rv, dist, params}`, `{:rv, dist, params, transform}`,
  `{:obs, target, value, meta}`, `{:meas_obs, target, value, op_info, meta}`, `{:det, fun, args}`
  (`builder.ex:34-82`, `node.ex`); params are `Nx` tensors or ref strings.

  An untyped `Nx.tensor(<float>)` is an f32 tensor in the reference: it travels as `{"f32": x}` so
  that the generator follows Nx's type inference for constants (`exponential.ex:16`, `normal.ex:19`).
  A `Custom` distribution's closure is Elixir code and does not travel: such a model is refused.
  """

  # one entry per module under lib/exmc/dist that the generator covers (exmc_amd/codegen.py _logpdf);
  # Exmc.Dist.Censored is `meta.censored` of an obs node, Exmc.Dist.Custom is a closure (refused)
  @dist %{
    Exmc.Dist.Normal => "normal",
    Exmc.Dist.HalfNormal => "half_normal",
    Exmc.Dist.HalfCauchy => "half_cauchy",
    Exmc.Dist.Cauchy => "cauchy",
    Exmc.Dist.Exponential => "exponential",
    Exmc.Dist.StudentT => "student_t",
    Exmc.Dist.Bernoulli => "bernoulli",
    Exmc.Dist.Beta => "beta",
    Exmc.Dist.Gamma => "gamma",
    Exmc.Dist.Laplace => "laplace",
    Exmc.Dist.Lognormal => "lognormal",
    Exmc.Dist.Poisson => "poisson",
    Exmc.Dist.Uniform01 => "uniform01",
    Exmc.Dist.Weibull => "weibull",
    Exmc.Dist.TruncatedNormal => "truncated_normal",
    Exmc.Dist.Mixture => "mixture",
    Exmc.Dist.GaussianRandomWalk => "gaussian_random_walk",
    Exmc.Dist.MvNormal => "mv_normal",
    Exmc.Dist.Dirichlet => "dirichlet"
  }

  @doc "The JSON document `exmc_amd.codegen` reads. `ncp`: whether the non-centred rewrite is wanted."
  def to_json(%Exmc.IR{nodes: nodes, data: data}, ncp \\ true) do
    %{
      "ncp" => ncp,
      "data" => data && Nx.to_list(data),
      # Map.keys/1 in THIS VM's iteration order = the order compiler.ex:176-180 (Map.values/1) builds
      # its terms in and sum_logps adds them: sorted for up to 32 keys, the map's internal hash order
      # above. A JSON object does not keep it, so it travels as a list and the generator follows it.
      "term_order" => Map.keys(nodes),
      "nodes" => Map.new(nodes, fn {id, %Exmc.Node{op: op}} -> {id, node(op)} end)
    }
    |> Jason.encode!()
  end

  defp node({:rv, dist, params}), do: node({:rv, dist, params, nil})

  defp node({:rv, dist, params, tr}) do
    %{
      "op" => "rv",
      "dist" => dist_name(dist),
      "transform" => tr && Atom.to_string(tr),
      "params" => Map.new(params, fn {k, v} -> {Atom.to_string(k), value(v)} end)
    }
  end

  defp node({:obs, target, value, meta}) do
    Map.merge(
      %{"op" => "obs", "target" => target, "value" => value(value)},
      Map.new(Map.take(meta, [:reduce, :weight, :mask, :censored, :likelihood]), fn {k, v} ->
        {Atom.to_string(k), value(v)}
      end)
    )
  end

  defp node({:meas_obs, target, value, {:affine, a, b}, _meta}) do
    %{"op" => "meas_obs", "target" => target, "value" => value(value), "info" => ["affine", value(a), value(b)]}
  end

  defp node({:det, fun, args}) do
    %{"op" => "det", "fun" => Atom.to_string(fun), "args" => Enum.map(args, &value/1)}
  end

  defp dist_name(dist) do
    case Map.fetch(@dist, dist) do
      {:ok, name} -> name
      :error -> raise ArgumentError, "#{inspect(dist)} has no generated form (a Custom closure does not travel as JSON)"
    end
  end

  defp value(%Nx.Tensor{} = t) do
    cond do
      Nx.shape(t) != {} -> Nx.to_list(t)
      Nx.type(t) == {:f, 32} -> %{"f32" => Nx.to_number(t)}
      true -> Nx.to_number(t)
    end
  end

  defp value(v) when is_atom(v) and not is_boolean(v) and not is_nil(v), do: Atom.to_string(v)
  defp value(v), do: v
end
