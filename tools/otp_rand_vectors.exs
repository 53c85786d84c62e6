# Pins the checker's restatement of OTP :rand (exsss + normal_s ziggurat) to a real BEAM.
# The build pipeline has no Erlang/OTP, so this is the one command a maintainer with OTP runs:
#
#     elixir tools/otp_rand_vectors.exs > tests/golden/otp_rand_vectors.txt
#     python -m pytest tests/test_otp_rand_vectors.py
#
# Each line: <seed> <kind> <index> <IEEE-754 bits of the float, hex>. Per seed the same stream is
# consumed the way the sampler does (sampler.ex:154, 343, 396, 897): 8 uniform_s then 8 normal_s
# from the state that follows, then 2000 more normal_s of which only those that left the
# ziggurat's fast path differ from a table lookup (all are printed; the test compares every one).
bits = fn x -> <<i::unsigned-64>> = <<x::float-64>>; Integer.to_string(i, 16) end

for seed <- [0, 1, 42, 7961, 123_456_789_012] do
  rng = :rand.seed_s(:exsss, seed)
  {rng, _} =
    Enum.reduce(0..7, {rng, nil}, fn i, {r, _} ->
      {u, r} = :rand.uniform_s(r)
      IO.puts("#{seed} uniform #{i} #{bits.(u)}")
      {r, nil}
    end)
  Enum.reduce(0..2007, rng, fn i, r ->
    {z, r} = :rand.normal_s(r)
    IO.puts("#{seed} normal #{i} #{bits.(z)}")
    r
  end)
end
