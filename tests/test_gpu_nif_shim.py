"""The NIF shims of c_src/ CALLED on the GPU (through tests/host/fake_erl_nif.c, there being no
BEAM here): `Elixir.Exmc.NUTS.NativeTree` runs the literal cases of the reference's own
test/native_tree_test.exs (tests/golden/reference_known_answers.json cites each) with the
reference's argument order and result shapes, and equals the Python mirror of the same C ABI bit
for bit; `Elixir.Exmc.NUTS.HipNative` equals exmc_amd.sampler on the same seeds."""
import json
import os

import numpy as np
import pytest

import nif_harness as H
from exmc_amd import models, native_tree, sampler

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden",
                                   "reference_known_answers.json")))["native_tree"]
A = np.array
GOLD_SV = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_traces.npz"))["sv_returns"]


@pytest.fixture(scope="module")
def mods(tmp_path_factory):
    return H.build(str(tmp_path_factory.mktemp("nif")))[1]


def test_init_get_roundtrip(hip, mods):
    """native_tree_test.exs:13-59: init_trajectory_bin + get_endpoint_bin round trip, the list twins,
    is_terminated false on a fresh trajectory."""
    nt, c = mods["NativeTree"], GOLD["init_get"]
    ref = nt.call("init_trajectory_bin", A(c["q"]), A(c["p"]), A(c["grad"]), c["logp"])
    for right in (True, False):
        q, p, g = nt.call("get_endpoint_bin", ref, right)
        assert (list(H.f64(q)), list(H.f64(p)), list(H.f64(g))) == (c["q"], c["p"], c["grad"])
    assert nt.call("is_terminated", ref) is False
    ref2 = nt.call("init_trajectory", c["q"], c["p"], c["grad"], c["logp"])
    assert nt.call("get_endpoint", ref2, True) == (c["q"], c["p"], c["grad"])
    r = nt.call("get_result", ref2)
    assert set(r) == {"q", "logp", "grad", "n_steps", "divergent", "accept_sum", "depth"}
    assert (r["q"], r["logp"], r["n_steps"], r["depth"], r["divergent"]) == (c["q"], c["logp"], 0, 0, False)


@pytest.mark.parametrize("name", ["depth0", "depth1", "divergent"])
def test_build_and_merge_cases(hip, mods, name):
    """native_tree_test.exs:63-178 through build_and_merge_bin/11 and the list twin build_and_merge/11;
    same bits as the batched Python mirror of the C ABI."""
    nt, c = mods["NativeTree"], GOLD[name]
    e = c["expect"]
    out = []
    for binary in (True, False):
        if binary:
            ref = nt.call("init_trajectory_bin", A(c["q"]), A(c["p"]), A(c["grad"]), c["logp"])
            assert nt.call("build_and_merge_bin", ref, A(c["all_q"]), A(c["all_p"]), A(c["all_logp"]),
                           A(c["all_grad"]), A(c["inv_mass"]), c["jlp0"], c["depth"], c["d"],
                           c["go_right"], c["seed"]) == H.Atom("ok")
            r = nt.call("get_result_bin", ref)
            assert set(r) == {"q_bin", "logp", "grad_bin", "n_steps", "divergent", "accept_sum", "depth"}
            r = dict(r, q=list(H.f64(r["q_bin"])), grad=list(H.f64(r["grad_bin"])))
        else:
            ref = nt.call("init_trajectory", c["q"], c["p"], c["grad"], c["logp"])
            assert nt.call("build_and_merge", ref, c["all_q"], c["all_p"], c["all_logp"], c["all_grad"],
                           c["inv_mass"], c["jlp0"], c["depth"], c["d"], c["go_right"],
                           c["seed"]) == H.Atom("ok")
            r = nt.call("get_result", ref)
        for k in ("n_steps", "depth", "divergent"):
            if k in e:
                assert r[k] == e[k], (name, k)
        if "terminated" in e:
            assert nt.call("is_terminated", ref) is e["terminated"]
        out.append((r["q"], r["grad"], r["logp"], r["n_steps"], r["depth"], r["divergent"], r["accept_sum"]))
    assert out[0] == out[1]
    # the same call through the Python mirror (one chain)
    d, n = c["d"], len(c["all_logp"])
    T = native_tree.Trajectories(A([c["q"]]), A([c["p"]]), A([c["grad"]]), [c["logp"]])
    T.build_and_merge_bin(A(c["all_q"]).reshape(1, n, d), A(c["all_p"]).reshape(1, n, d),
                          A([c["all_logp"]]), A(c["all_grad"]).reshape(1, n, d), c["inv_mass"],
                          [c["jlp0"]], c["depth"], d, c["go_right"], [c["seed"]])
    m = T.get_result_bin()
    assert list(m["q_bin"][0]) == out[0][0] and m["logp"][0] == out[0][2]
    assert (m["n_steps"][0], m["depth"][0], bool(m["divergent"][0])) == out[0][3:6]
    assert m["accept_sum"][0] == out[0][6]


def test_build_subtree_bin_record(hip, mods):
    """build_subtree_bin/10 returns the sixteen keys of lib.rs:146-210; values equal the mirror's."""
    nt, c = mods["NativeTree"], GOLD["depth1"]
    r = nt.call("build_subtree_bin", A(c["all_q"]), A(c["all_p"]), A(c["all_logp"]), A(c["all_grad"]),
                A(c["inv_mass"]), c["jlp0"], c["depth"], c["d"], c["go_right"], c["seed"])
    assert set(r) == {"q_left_bin", "p_left_bin", "grad_left_bin", "q_right_bin", "p_right_bin",
                      "grad_right_bin", "q_prop_bin", "logp_prop", "grad_prop_bin", "log_sum_weight",
                      "n_steps", "divergent", "accept_sum", "turning", "depth", "rho_bin"}
    d, n = c["d"], len(c["all_logp"])
    m = native_tree.build_subtree_bin(A(c["all_q"]).reshape(1, n, d), A(c["all_p"]).reshape(1, n, d),
                                      A([c["all_logp"]]), A(c["all_grad"]).reshape(1, n, d),
                                      c["inv_mass"], [c["jlp0"]], c["depth"], d, c["go_right"], [c["seed"]])
    for k in ("q_left_bin", "p_right_bin", "q_prop_bin", "rho_bin", "grad_prop_bin"):
        assert np.array_equal(H.f64(r[k]), m[k][0]), k
    for k in ("logp_prop", "log_sum_weight", "accept_sum"):
        assert r[k] == m[k][0], k
    assert (r["n_steps"], r["depth"], r["divergent"], r["turning"]) == \
        (m["n_steps"][0], m["depth"][0], bool(m["divergent"][0]), bool(m["turning"][0]))
    assert r["n_steps"] == 2 and r["depth"] == 1


def test_build_full_tree_bin_cases(hip, mods):
    """native_tree_test.exs:182-291 with the reference's seventeen arguments."""
    nt = mods["NativeTree"]
    c = GOLD["full_tree"]
    args = [A(c["q0"]), A(c["p0"]), A(c["grad0"]), c["logp0"], A(c["fwd_q"]), A(c["fwd_p"]),
            A(c["fwd_logp"]), A(c["fwd_grad"]), A(c["bwd_q"]), A(c["bwd_p"]), A(c["bwd_logp"]),
            A(c["bwd_grad"]), A(c["inv_mass"]), c["jlp0"], c["max_depth"], c["d"], c["seed"]]
    r = nt.call("build_full_tree_bin", *args)
    assert set(r) == {"q_bin", "logp", "grad_bin", "n_steps", "divergent", "accept_sum", "depth"}
    e = c["expect"]
    assert r["n_steps"] > e["n_steps_gt"] and r["accept_sum"] > e["accept_sum_gt"]
    assert e["depth_gt"] < r["depth"] <= e["depth_le"]
    k = lambda name: A(c[name])[None, :, None]  # noqa: E731
    m = native_tree.build_full_tree_bin(
        A([c["q0"]]), A([c["p0"]]), A([c["grad0"]]), [c["logp0"]], k("fwd_q"), k("fwd_p"),
        A([c["fwd_logp"]]), k("fwd_grad"), k("bwd_q"), k("bwd_p"), A([c["bwd_logp"]]), k("bwd_grad"),
        c["inv_mass"], [c["jlp0"]], c["max_depth"], 1, [c["seed"]])
    assert np.array_equal(H.f64(r["q_bin"]), m["q_bin"][0]) and r["logp"] == m["logp"][0]
    assert (r["n_steps"], r["depth"], r["divergent"], r["accept_sum"]) == \
        (m["n_steps"][0], m["depth"][0], bool(m["divergent"][0]), m["accept_sum"][0])
    c = GOLD["full_tree_divergent"]
    n = c["n"]
    full = lambda v: np.full(n, float(v))  # noqa: E731
    r = nt.call("build_full_tree_bin", A(c["q0"]), A(c["p0"]), A(c["grad0"]), c["logp0"],
                full(c["fwd_q_value"]), full(c["p_value"]), full(c["logp_value"]), full(c["grad_value"]),
                full(c["bwd_q_value"]), full(c["p_value"]), full(c["logp_value"]), full(c["grad_value"]),
                A(c["inv_mass"]), c["jlp0"], c["max_depth"], c["d"], c["seed"])
    assert r["divergent"] is True and r["n_steps"] <= c["expect"]["n_steps_le"]


def sampler_rows_after(spec, num_warmup, seed, skip, n):
    """Rows [skip, skip + n) of the chain stream_begin(nil, num_warmup, 10, 0.8, seed) starts, through
    the Python mirror's chunked stream (the same C ABI)."""
    import ctypes as C
    from exmc_amd import _lib
    comp = sampler.compile(spec)
    tun = _lib.Tuning()
    comp.check(comp.L.exmc_hip_stream_begin(comp.h, None, sampler._c_opts(sampler._merge_opts(
        dict(num_warmup=num_warmup, seed=seed, max_tree_depth=10, target_accept=0.8))), C.byref(tun)))
    t, tr = sampler._host_trace(1, skip + n, spec.d)
    dv = C.c_int32()
    comp.check(comp.L.exmc_hip_stream_next_host(comp.h, skip + n, tr, C.byref(dv)))
    return {k: v[:, skip:] for k, v in t.items()}, tun


def test_hip_native_equals_the_python_mirror(hip, mods):
    """model_create -> warmup -> sample_chains / sample / stream through the NIF functions."""
    hn = mods["HipNative"]
    spec = models.eight_schools()
    ok, ref = hn.call("model_create", spec.kind, spec.data)
    assert ok == H.Atom("ok")
    assert hn.call("model_set_flat_order", ref, spec.flat_order()) == H.Atom("ok")
    q0 = spec.to_unconstrained(spec.default_init)
    tun = hn.call("warmup", ref, q0, 120, 10, 0.8, 42)
    comp = sampler.compile(spec)
    opts = dict(num_warmup=120, num_samples=30, seed=42)
    t2 = sampler.warmup(comp, spec.default_init, opts)
    assert tun["epsilon"] == t2["epsilon"] and np.array_equal(H.f64(tun["inv_mass"]), t2["inv_mass"])
    tr, lf, dv = hn.call("sample_chains", ref, tun["epsilon"], H.f64(tun["inv_mass"]), q0, 6, 0, 6, 30, 10, 42)
    _, _, extra = sampler.sample_compiled_tuned(comp, t2, spec.default_init, opts, num_chains=6)
    raw = extra["raw"]
    assert np.array_equal(H.f64(tr["draws"]).reshape(6, 30, spec.d), raw["draws"])
    assert np.array_equal(H.i32(tr["tree_depth"]).reshape(6, 30), raw["tree_depth"])
    assert np.array_equal(H.i32(tr["n_steps"]).reshape(6, 30), raw["n_steps"])
    assert np.array_equal(H.f64(tr["energy"]).reshape(6, 30), raw["energy"])
    assert lf == extra["total_leapfrogs"]
    # logp_grad and multi_step keep the B2 shapes
    lp, g = hn.call("logp_grad", ref, np.tile(q0, 3), 3)
    assert H.f64(lp).shape == (3,) and H.f64(g).shape == (30,)
    aq, ap, al, ag = hn.call("multi_step", ref, q0, np.ones(spec.d), H.f64(g)[:spec.d], 0.1,
                             np.ones(spec.d), 4, 1)
    assert H.f64(aq).shape == (4 * spec.d,) and H.f64(al).shape == (4,)
    # random init (nil) and the streaming pair
    tun3 = hn.call("stream_begin", ref, None, 50, 10, 0.8, 3)
    rows, _ = hn.call("stream_next", ref, 5)
    assert H.f64(rows["draws"]).shape == (5 * spec.d,) and tun3["epsilon"] > 0
    with pytest.raises(H.BadArg):
        hn.call("warmup", ref, np.zeros(3), 10, 10, 0.8, 1)
    # the sender of sample_stream/4: one launch, a message per finished draw from a thread of the
    # library (enif_send), then {:exmc_done, n, divergences}; the rows continue the resident chain
    more, _ = sampler_rows_after(spec, 50, 3, 5, 12)
    assert hn.call("stream_run", ref, 12, 1) == H.Atom("ok")          # 1 = the harness's one process
    msgs = H.mailbox(hn, 13)
    assert msgs[-1][0] == H.Atom("exmc_done") and msgs[-1][1] == 12
    for i, (tag, idx, qb, st) in enumerate(msgs[:-1]):
        assert tag == H.Atom("exmc_sample") and idx == i + 1 and len(st) == 5
        assert np.array_equal(H.f64(qb), more["draws"][0][i])
        assert st[1] == int(more["n_steps"][0][i]) and st[4] == float(more["energy"][0][i])
    with pytest.raises(H.BadArg):
        hn.call("stream_run", ref, 0, 1)
    # warm start and the dense mass through the shim = the Python mirror of the same C ABI
    ws = hn.call("warmup_from", ref, q0, 200, 10, 0.8, 7, tun["epsilon"], H.f64(tun["inv_mass"]))
    _, st = sampler.sample(spec, spec.default_init, dict(num_warmup=200, num_samples=5, seed=7, lanes_per_chain=16,
                                                          warm_start=dict(step_size=tun["epsilon"],
                                                                          inv_mass_diag=H.f64(tun["inv_mass"]))))
    assert ws["epsilon"] == st["step_size"] and np.array_equal(H.f64(ws["inv_mass"]), st["inv_mass_diag"])
    dn = hn.call("warmup_dense", ref, q0, 300, 10, 0.8, 13, 16)
    d2 = sampler.warmup(comp, spec.default_init, dict(num_warmup=300, seed=13, lanes_per_chain=16, dense_mass=True))
    assert dn["epsilon"] == d2["epsilon"]
    assert np.array_equal(H.f64(dn["cov"]).reshape(spec.d, spec.d), d2["cov"])
    assert np.array_equal(H.f64(dn["chol_cov"]).reshape(spec.d, spec.d), d2["chol_cov"])
    assert hn.call("clear_dense_mass", ref) == H.Atom("ok")
    assert hn.call("set_dense_mass", ref, H.f64(dn["cov"]), H.f64(dn["chol_cov"])) == H.Atom("ok")
    assert hn.call("clear_dense_mass", ref) == H.Atom("ok")
    with pytest.raises(H.BadArg):
        hn.call("set_dense_mass", ref, np.zeros(5), np.zeros(5))


def test_sample_with_warm_start_and_dense_mass_through_the_nif(hip, mods):
    """HipNative.sample_warm/9 and sample_dense/8 (round 6: sample/3 with opts[:warm_start], sampler.ex:167-197, and
    dense_mass: true, sampler.ex:156) = the Python mirror of the same C entry points (which the checker pins),
    draws included; on sv, whose kernel order is not the flat order."""
    hn = mods["HipNative"]
    for spec, lanes in ((models.eight_schools(), 16), (models.sv(GOLD_SV), 64)):
        ok, ref = hn.call("model_create", spec.kind, spec.data)
        assert ok == H.Atom("ok")
        assert hn.call("model_set_flat_order", ref, spec.flat_order()) == H.Atom("ok")
        q0 = spec.to_unconstrained(spec.default_init)
        nw, ns = (150, 20) if spec.d < 50 else (60, 6)
        tr0, tun0, dv0 = hn.call("sample", ref, q0, nw, ns, 10, 0.8, 5)
        t0, s0 = sampler.sample(spec, spec.default_init, dict(num_warmup=nw, num_samples=ns, seed=5))
        assert tun0["epsilon"] == s0["step_size"]
        assert np.array_equal(H.f64(tr0["draws"]).reshape(ns, spec.d), s0["raw"]["draws"][0])
        # warm start from that run's tuning: min(num_warmup, 50) iterations, then the same chain's draws
        trw, tunw, dvw = hn.call("sample_warm", ref, q0, nw, ns, 10, 0.8, 9, tun0["epsilon"], H.f64(tun0["inv_mass"]))
        tw, sw = sampler.sample(spec, spec.default_init, dict(num_warmup=nw, num_samples=ns, seed=9,
                                                              warm_start=dict(step_size=s0["step_size"],
                                                                              inv_mass_diag=s0["inv_mass_diag"])))
        assert tunw["epsilon"] == sw["step_size"] and np.array_equal(H.f64(tunw["inv_mass"]), sw["inv_mass_diag"])
        assert np.array_equal(H.f64(trw["draws"]).reshape(ns, spec.d), sw["raw"]["draws"][0])
        assert np.array_equal(H.i32(trw["n_steps"]), sw["raw"]["n_steps"][0]) and dvw == sw["divergences"]
        with pytest.raises(H.BadArg):
            hn.call("sample_warm", ref, q0, nw, ns, 10, 0.8, 9, tun0["epsilon"], np.zeros(3))
        if spec.d > 50:
            continue      # (the dense windows of sv at this size are a test of their own, test_dense_mass_*)
        trd, tund, dvd = hn.call("sample_dense", ref, q0, 300, ns, 10, 0.8, 13, 0)
        td, sd = sampler.sample(spec, spec.default_init, dict(num_warmup=300, num_samples=ns, seed=13, dense_mass=True))
        assert tund["epsilon"] == sd["step_size"]
        assert np.array_equal(H.f64(tund["chol_cov"]).reshape(spec.d, spec.d), sd["chol_cov"])
        assert np.array_equal(H.f64(tund["cov"]).reshape(spec.d, spec.d), sd["cov"])
        assert np.array_equal(H.f64(trd["draws"]).reshape(ns, spec.d), sd["raw"]["draws"][0])
        assert dvd == sd["divergences"]


def test_sample_independent_through_the_nif(hip, mods):
    """HipNative.sample_independent/10 (sample_chains vectorized: false, sampler.ex:1139-1176) = the Python
    mirror of the same C entry point: a shard [2, 7) of 9 chains, every chain its own adaptation."""
    hn = mods["HipNative"]
    spec = models.eight_schools()
    ok, ref = hn.call("model_create", spec.kind, spec.data)
    assert ok == H.Atom("ok")
    assert hn.call("model_set_flat_order", ref, spec.flat_order()) == H.Atom("ok")
    q0 = spec.to_unconstrained(spec.default_init)
    tr, tune, lf, dv = hn.call("sample_independent", ref, q0, 9, 2, 7, 80, 25, 10, 0.8, 17)
    comp = sampler.compile(spec)
    try:
        opts = dict(num_warmup=80, num_samples=25, seed=17, vectorized=False, init_values=spec.default_init)
        _, stats = sampler.sample_chains_independent_compiled(comp, 9, opts, chain_lo=2, chain_hi=7)
    finally:
        comp.close()
    raw = stats[0]["extra"]["raw"]
    assert np.array_equal(H.f64(tr["draws"]).reshape(5, 25, spec.d), raw["draws"])
    assert np.array_equal(H.i32(tr["n_steps"]).reshape(5, 25), raw["n_steps"])
    assert np.array_equal(H.f64(tr["energy"]).reshape(5, 25), raw["energy"])
    tune = H.f64(tune).reshape(5, 3 + spec.d)
    assert np.array_equal(tune, stats[0]["extra"]["tuning"])      # step size, warmup divergences / leapfrogs, inv_mass
    assert [float(t[0]) for t in tune] == [s_["step_size"] for s_ in stats]
    assert len({float(t[0]) for t in tune}) == 5                  # every chain tuned its own step size
    assert lf == int(raw["n_steps"].sum())
    with pytest.raises(H.BadArg):
        hn.call("sample_independent", ref, q0, 9, 7, 7, 80, 25, 10, 0.8, 17)   # an empty chain range


def test_generated_model_through_the_nif(hip, mods):
    """HipNative.model_create_plugin/2: a model generated from Builder IR (eight schools as 26 nodes,
    and sv in the lane layout) reaches the BEAM side -- the plug-in library is dlopen'ed by the shim,
    warmup / sample_chains / stream_run run through ITS entry points and give what the Python mirror
    gives through the same library."""
    from exmc_amd import codegen as cg
    import gen_models as GM
    hn = mods["HipNative"]
    cases = []
    init = {n: 0.0 for n in ["mu"] + ["theta_%d" % j for j in range(8)]}
    init["tau"] = 1.0
    cases.append(cg.compile_ir(cg.eight_schools_ir(), name="gen_eight_schools", default_init=init))
    ir, ncp, hand, lanes = GM.baseline_pair("sv")
    cases.append(cg.compile_ir(ir, ncp=ncp, name="gen_sv", default_init=hand.default_init, lanes=lanes))
    for spec in cases:
        ok, ref = hn.call("model_create_plugin", spec.lib_path, spec.data)
        assert ok == H.Atom("ok")
        nw, ns, nc = (100, 20, 5)
        q0 = spec.to_unconstrained(spec.default_init)
        tun = hn.call("warmup", ref, q0, nw, 10, 0.8, 42)
        comp = sampler.compile(spec)
        opts = dict(num_warmup=nw, num_samples=ns, seed=42)
        t2 = sampler.warmup(comp, spec.default_init, opts)
        assert tun["epsilon"] == t2["epsilon"] and np.array_equal(H.f64(tun["inv_mass"]), t2["inv_mass"])
        tr, lf, dv = hn.call("sample_chains", ref, tun["epsilon"], H.f64(tun["inv_mass"]), q0, nc, 0, nc, ns, 10, 42)
        _, _, extra = sampler.sample_compiled_tuned(comp, t2, spec.default_init, opts, num_chains=nc)
        raw = extra["raw"]
        assert np.array_equal(H.f64(tr["draws"]).reshape(nc, ns, spec.d), raw["draws"])
        assert np.array_equal(H.i32(tr["n_steps"]).reshape(nc, ns), raw["n_steps"])
        assert lf == extra["total_leapfrogs"]
        # a kind the plug-in does not carry is its own error, through its own last_error
        bad = hn.call("model_create_plugin", spec.lib_path, np.zeros(spec.data.size + 3))
        assert bad[0] == H.Atom("error") and "data length" in bad[1]
