"""Builder IR -> generated value+gradient (exmc_amd/codegen.py), checked on the CPU: the generated
text compiled with gcc (tests/gen_checker.py) against the hand-written oracle models, against
central differences, and the plug-in build of the HIP library (cross-compiled, not run)."""
import ctypes as C
import os

import numpy as np
import pytest

import gen_checker as GC
import gen_models as GM
import oracle as O
from exmc_amd import _lib, codegen as cg, models

DET = O.Cfg(1, 1)


def test_simple_matches_handwritten_oracle_model():
    """Forward value: same Nx operation sequence as the hand restatement -> same bits. Gradient:
    the generator's reverse-mode rules vs the hand-derived form -> a few ulp."""
    gen = cg.generate(GM.simple_ir())
    assert gen.d == 2 and gen.var_names == ["mu", "sigma"] and gen.transforms == {"sigma": "log"}
    m, mh = GC.model(gen, 1), O.simple()
    rng = np.random.default_rng(0)
    for i in range(50):
        q = rng.normal(size=2) * (1.0 + i % 3)
        lp, g = m.logp_grad(q, DET)
        lph, gh = mh.logp_grad(q, DET)
        assert lp == lph
        np.testing.assert_allclose(g, gh, rtol=1e-13, atol=1e-13)
    # outside the :log clamp (transform.ex:17-29) the gradient of sigma's coordinate is cut
    for z in (250.0, -250.0):
        lp, g = m.logp_grad(np.array([0.3, z]), DET)
        lph, gh = mh.logp_grad(np.array([0.3, z]), DET)
        assert lp == lph and g[1] == 0.0 and gh[1] == 0.0


def test_eight_schools_builder_ir_matches_handwritten_gradient():
    """26 Builder nodes, non-centred rewrite applied: the same posterior as the posteriordb
    script's Custom-dist model up to its dropped 8*0.5*log(2pi) constant."""
    gen = cg.generate(GM.eight_schools_ir())
    assert gen.d == 10 and sorted(gen.ncp_info) == ["theta_%d" % j for j in range(8)]
    m, mh = GC.model(gen, 1), O.eight_schools()
    rng = np.random.default_rng(1)
    diffs = []
    for _ in range(20):
        q = rng.normal(size=10)
        lp, g = m.logp_grad(q, DET)
        lph, gh = mh.logp_grad(q, DET)
        diffs.append(lp - lph)
        np.testing.assert_allclose(g, gh, rtol=1e-12, atol=1e-12)
    assert np.ptp(diffs) < 1e-12
    assert abs(diffs[0] + 8 * 0.5 * cg.LOG_2PI_F32) < 1e-12
    # centred variant (ncp: false, rewrite.ex:24-27) is a different parameterisation
    genc = cg.generate(GM.eight_schools_ir(), ncp=False)
    assert genc.ncp_info == {} and genc.digest != gen.digest


def test_every_distribution_and_transform_against_central_differences():
    gen = cg.generate(GM.zoo_ir())
    assert gen.d == 9
    m = GC.model(gen, 1)
    rng = np.random.default_rng(2)
    for _ in range(10):
        q = rng.normal(size=gen.d) * 0.8
        lp, g = m.logp_grad(q, DET)
        assert np.isfinite(lp)
        fd = np.zeros(gen.d)
        for i in range(gen.d):
            e = np.zeros(gen.d)
            e[i] = 1e-6
            fd[i] = (m.logp_grad(q + e, DET)[0] - m.logp_grad(q - e, DET)[0]) / 2e-6
        np.testing.assert_allclose(g, fd, rtol=2e-6, atol=2e-6)


def test_distribution_terms_match_oracle_known_answer_functions():
    """Each generated logpdf equals the oracle's restatement of lib/exmc/dist/<name>.ex (which the
    golden doctest literals pin, tests/test_golden.py) bit for bit in deterministic-math mode."""
    L = O.lib()
    cases = [
        ("normal", dict(mu=0.3, sigma=1.7), lambda x: L.exo_dist_normal(x, 0.3, 1.7, 1)),
        ("half_normal", dict(sigma=1.7), lambda x: L.exo_dist_half_normal(abs(x), 1.7, 1)),
        ("half_cauchy", dict(scale=2.5), lambda x: L.exo_dist_half_cauchy(abs(x), 2.5, 1)),
        ("exponential", {"lambda": 0.7}, lambda x: L.exo_dist_exponential(abs(x), 0.7, 1)),
        ("student_t", dict(df=4.5, loc=0.2, scale=1.1),
         lambda x: L.exo_dist_student_t(x, 4.5, 0.2, 1.1, 1)),
    ]
    rng = np.random.default_rng(3)
    for dist, params, ref in cases:
        positive = dist in ("half_normal", "half_cauchy", "exponential")
        ir = cg.IR()
        # a free "x" with prior dist(params); its term is the only one
        ir.rv("x", dist, params)
        gen = cg.generate(ir)
        for _ in range(20):
            x = float(rng.normal()) * 2.0
            x = abs(x) if positive else x
            lp, _ = GC.logp_grad(gen, np.array([x]))
            assert lp == ref(x), (dist, x, lp, ref(x))
    # bernoulli as an observation of p = sigmoid(z)
    ir = cg.IR()
    ir.rv("p", "normal", dict(mu=0.0, sigma=10.0), transform="logit")
    ir.rv("yy", "bernoulli", dict(p="p"))
    ir.obs("y_obs", "yy", 1.0)
    gen = cg.generate(ir)
    lp, g = GC.logp_grad(gen, np.array([0.4]))
    assert np.isfinite(lp) and g[0] != 0.0


def test_generation_is_deterministic_and_rejects_what_it_does_not_cover():
    a, b = cg.generate(GM.zoo_ir()), cg.generate(GM.zoo_ir())
    assert a.header == b.header and a.digest == b.digest
    with pytest.raises(cg.CodegenError):
        cg.generate(cg.IR().rv("x", "weibull", dict(k=1.0)))
    with pytest.raises(cg.CodegenError):
        cg.generate(cg.IR().rv("x", "normal", dict(mu=0.0)))            # missing param
    with pytest.raises(cg.CodegenError):
        cg.generate(cg.IR().rv("x", "normal", dict(mu="nope", sigma=1.0)))
    with pytest.raises(cg.CodegenError):
        cg.generate(cg.IR().rv("x", "normal", dict(mu=0.0, sigma=1.0), transform="stick_breaking"))
    ir = cg.IR()                               # above the one-lane limit: the lane layout (round 3)
    for i in range(cg.MAX_D + 1):
        ir.rv("x%02d" % i, "normal", dict(mu=0.0, sigma=1.0))
    assert cg.generate(ir).lanes == 16
    with pytest.raises(cg.CodegenError):
        ir = cg.IR()
        for i in range(cg.MAX_D_LANES + 1):
            ir.rv("x%03d" % i, "normal", dict(mu=0.0, sigma=1.0))
        cg.generate(ir)
    with pytest.raises(cg.CodegenError):
        cg.IR().rv("x", "normal", dict(mu=0.0, sigma=1.0)).obs("o", "x", [1.0], censored="upper")
    with pytest.raises(cg.CodegenError):       # only observed nodes: nothing to sample
        cg.generate(cg.IR().rv("x", "normal", dict(mu=0.0, sigma=1.0)).obs("o", "x", 1.0))


def test_spec_init_and_trace_follow_the_non_centred_rewrite():
    """invert_ncp_init (sampler.ex:362-386) and reconstruct_ncp (sampler.ex:1300-1313)."""
    gen = cg.generate(GM.eight_schools_ir())
    spec = cg.GeneratedSpec(gen, lib_path=None)
    init = dict(mu=1.0, tau=2.0, **{"theta_%d" % j: 1.0 + 2.0 * (j - 3) for j in range(8)})
    q = spec.to_unconstrained(init)
    assert q[0] == 1.0 and q[1] == np.log(2.0)
    np.testing.assert_allclose(q[2:], np.arange(8) - 3.0)
    x = spec.constrain(q[None, :])[0]
    np.testing.assert_allclose(x, [init[n] for n in spec.var_names], rtol=1e-15)


def test_plugin_library_builds_and_exports_the_c_abi():
    """hipcc cross-compiles the NUTS kernels around the generated functor; the plug-in carries
    every symbol of include/exmc_hip.h (no compute without a GPU)."""
    gen = cg.generate(GM.simple_ir())
    so = cg.build_plugin(gen)
    assert os.path.exists(so)
    L = C.CDLL(so)
    for name in _lib.EXPORTS:
        getattr(L, name)


def test_json_front_door(tmp_path):
    """python -m exmc_amd.codegen model.json out_dir: the same generator for a non-Python host."""
    import json
    doc = {"ncp": True, "nodes": {
        "mu": {"op": "rv", "dist": "normal", "params": {"mu": 0.0, "sigma": 5.0}},
        "sigma": {"op": "rv", "dist": "exponential", "params": {"lambda": 1.0}, "transform": "log"},
        "y": {"op": "rv", "dist": "normal", "params": {"mu": "mu", "sigma": "sigma"}},
        "y_obs": {"op": "obs", "target": "y", "value": list(models.SIMPLE_Y)}}}
    src = tmp_path / "m.json"
    src.write_text(json.dumps(doc))
    cg.main([str(src), str(tmp_path / "out"), "--no-build"])
    meta = json.loads((tmp_path / "out" / "model.json").read_text())
    ref = cg.generate(GM.simple_ir())
    assert meta["digest"] == ref.digest and meta["d"] == 2 and meta["var_names"] == ["mu", "sigma"]
    assert (tmp_path / "out" / "exmc_gen_model.h").read_text() == ref.header
    with pytest.raises(cg.CodegenError):
        cg.ir_from_json({"nodes": {"x": {"op": "data"}}})
    # "rewrite": the reference's passes run first -- the same model written without its transform
    del doc["nodes"]["sigma"]["transform"]
    doc["rewrite"] = True
    doc["nodes"]["shift"] = {"op": "det", "fun": "affine", "args": [1.0, 0.0, "mu"]}
    src.write_text(json.dumps(doc))
    cg.main([str(src), str(tmp_path / "out2"), "--no-build"])
    meta2 = json.loads((tmp_path / "out2" / "model.json").read_text())
    assert meta2["transforms"] == {"sigma": "log"} and meta2["d"] == 2


def test_plates_across_lanes_agree_with_the_one_lane_form():
    """codegen_vec: families of identical terms run on different lanes with per-lane constants
    (eight schools: the nine Normal priors and the eight likelihood terms; shared mu, tau reduced
    in one butterfly). Same value and gradient as the scalar form up to the summation order."""
    from exmc_amd import codegen_vec
    gen = cg.generate(GM.eight_schools_ir())
    assert gen.lanes == 16 and gen.vec["n_families"] == 2 and gen.vec["n_reduced"] == 3
    assert cg.generate(GM.eight_schools_ir(), vectorize=False).lanes == 1
    rng = np.random.default_rng(4)
    for ir in (GM.eight_schools_ir(), GM.simple_ir(), GM.zoo_ir()):
        g2 = cg.generate(ir)
        assert g2.vec is not None
        for _ in range(25):
            q = rng.normal(size=g2.d) * 1.3
            a, ga = GC.logp_grad(g2, q, 1)
            b, gb = GC.logp_grad(g2, q, 16)
            assert abs(a - b) <= 1e-13 * max(1.0, abs(a))
            np.testing.assert_allclose(ga, gb, rtol=1e-12, atol=1e-12)
    # nothing to spread: a single term per shape, or more dimensions than lanes
    ir = cg.IR().rv("x", "normal", dict(mu=0.0, sigma=1.0)).rv("s", "half_cauchy", dict(scale=1.0),
                                                             transform="log")
    assert codegen_vec.generate(ir) is None and cg.generate(ir).lanes == 1
    big = cg.IR()
    for i in range(18):
        big.rv("x%02d" % i, "normal", dict(mu=0.0, sigma=1.0))
    assert cg.generate(big).lanes == 1
    # an observation vector longer than the lane group is taken 16 elements at a time
    y = rng.normal(size=40) * 2.0 + 1.0
    g3 = cg.generate(GM.simple_ir(y=y))
    assert g3.lanes == 16 and g3.vec["n_families"] == 3
    for _ in range(10):
        q = rng.normal(size=2)
        a, ga = GC.logp_grad(g3, q, 1)
        b, gb = GC.logp_grad(g3, q, 16)
        assert abs(a - b) <= 1e-13 * max(1.0, abs(a))
        np.testing.assert_allclose(ga, gb, rtol=1e-12, atol=1e-12)
