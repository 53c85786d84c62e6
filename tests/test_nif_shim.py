"""The drop-in boundary of SURVEY 8(b)(i): c_src/exmc_native_tree_nif.c exports an ErlNifEntry named
`Elixir.Exmc.NUTS.NativeTree` with the functions of native/exmc_tree/src/lib.rs:37-442 at the
same arities (the Elixir stubs lib/exmc/nuts/native_tree.ex:20-110 are what the BEAM binds them
to), and c_src/exmc_hip_nif.c exports `Elixir.Exmc.NUTS.HipNative`. Compiled here against
c_src/erl_nif_decl.h (there is no OTP in this image) and called through tests/host/fake_erl_nif.c.
Without a GPU the calls must fail the way the product fails: no CPU fallback."""
import os
import subprocess

import numpy as np
import pytest

import nif_harness as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (name, arity, dirty) of every #[rustler::nif] in native/exmc_tree/src/lib.rs, in file order
NATIVE_TREE = [("init_trajectory_bin", 4, True), ("is_terminated", 1, False),
               ("get_endpoint_bin", 2, False), ("build_and_merge_bin", 11, True),
               ("build_subtree_bin", 10, True), ("build_full_tree_bin", 17, True),
               ("get_result_bin", 1, False), ("init_trajectory", 4, True), ("get_endpoint", 2, False),
               ("build_and_merge", 11, True), ("get_result", 1, False)]


@pytest.fixture(scope="module")
def mods(tmp_path_factory):
    return H.build(str(tmp_path_factory.mktemp("nif")))[1]


def test_shims_compile_as_plain_c_with_warnings_as_errors(tmp_path):
    for src in ("exmc_native_tree_nif.c", "exmc_hip_nif.c"):
        subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-fPIC", "-c", "-o",
                               str(tmp_path / (src + ".o")), os.path.join(ROOT, "c_src", src)])


def test_native_tree_entry_matches_the_rust_crate(mods):
    m = mods["NativeTree"]
    assert m.name == "Elixir.Exmc.NUTS.NativeTree"          # rustler::init!, lib.rs:442
    e = m.entry.contents
    assert (e.major, e.vm_variant) == (2, b"beam.vanilla") and e.load
    table = m.table()
    assert [(n, a) for n, a, _ in table] == [(n, a) for n, a, _ in NATIVE_TREE]
    for (name, _, flags), (_, _, dirty) in zip(table, NATIVE_TREE):
        # what the crate schedules "DirtyCpu" is dirty here too; its plain accessors (memory reads in
        # Rust) are blocking device copies in this shim, so they are dirty IO-bound jobs as well
        assert flags != 0, name
        assert dirty in (True, False)


def test_native_tree_entry_matches_the_elixir_stubs(mods):
    """Every `def name(args), do: :erlang.nif_error(:nif_not_loaded)` of native_tree.ex, transcribed
    as (name, arity): the functions the BEAM will look for when it loads the module."""
    stubs = {"init_trajectory_bin": 4, "get_endpoint_bin": 2, "build_and_merge_bin": 11,
             "build_subtree_bin": 10, "build_full_tree_bin": 17, "get_result_bin": 1,
             "init_trajectory": 4, "is_terminated": 1, "get_endpoint": 2, "build_and_merge": 11,
             "get_result": 1}
    assert {n: a for n, a, _ in mods["NativeTree"].table()} == stubs


def test_hip_native_entry(mods):
    m = mods["HipNative"]
    assert m.name == "Elixir.Exmc.NUTS.HipNative"
    assert {(n, a) for n, a, _ in m.table()} == {
        ("model_create", 2), ("model_create_plugin", 2), ("model_set_flat_order", 2), ("logp_grad", 3), ("multi_step", 8),
        ("leapfrog_chain_normal", 7),
        ("warmup", 6), ("warmup_from", 8), ("warmup_dense", 7), ("set_dense_mass", 3), ("clear_dense_mass", 1),
        ("sample_chains", 10), ("sample_independent", 10), ("sample", 7), ("sample_warm", 9), ("sample_dense", 8),
        ("stream_begin", 6), ("stream_next", 2), ("stream_run", 3)}
    assert all(flags != 0 for _, _, flags in m.table())      # every call waits on the GPU


def _no_gpu():
    from exmc_amd import _lib
    return _lib.load().exmc_hip_device_count() == 0


def test_decode_failures_are_badarg(mods):
    nt = mods["NativeTree"]
    q = np.zeros(3)
    with pytest.raises(H.BadArg):
        nt.call("init_trajectory_bin", q, np.zeros(2), q, 0.0)          # p has another length
    with pytest.raises(H.BadArg):
        nt.call("init_trajectory_bin", q, q, q, H.Atom("zero"))          # logp is not a number
    with pytest.raises(H.BadArg):
        nt.call("init_trajectory_bin", b"\x00" * 7, q, q, 0.0)          # not a whole number of f64
    with pytest.raises(H.BadArg):
        nt.call("is_terminated", 5)                                       # not a resource
    with pytest.raises(H.BadArg):
        nt.call("build_subtree_bin", np.zeros(4), np.zeros(4), np.zeros(1), np.zeros(4), np.ones(2),
                0.0, 1, 2, True, 7)                                       # depth 1 needs 2 states
    with pytest.raises(AttributeError):
        nt.call("init_trajectory_bin", q, q, q)                          # arity 3 is not exported
    hn = mods["HipNative"]
    with pytest.raises(H.BadArg):
        hn.call("model_create", "eight_schools", np.zeros(16))


def test_plugin_loading_binds_the_plugins_own_abi(mods):
    """model_create_plugin dlopens the library a generated model was compiled into and binds its
    entry points into a table of the handle's own. Without a GPU the plug-in's model_create answers
    (its own message, no CPU fallback); a library that is not a plug-in and a missing file are errors."""
    from exmc_amd import _lib, codegen as cg
    hn = mods["HipNative"]
    r = hn.call("model_create_plugin", "/nonexistent/libexmc_hip_gen.so", np.zeros(1))
    assert r[0] == H.Atom("error") and "nonexistent" in r[1]
    libm = "/lib/x86_64-linux-gnu/libm.so.6"
    if os.path.exists(libm):
        r = hn.call("model_create_plugin", libm, np.zeros(1))
        assert r[0] == H.Atom("error") and r[1].startswith("exmc_hip_")       # the first symbol it lacks
    with pytest.raises(H.BadArg):
        hn.call("model_create_plugin", 5, np.zeros(1))
    gen = cg.generate(cg.eight_schools_ir())
    so = cg.build_plugin(gen)
    r = hn.call("model_create_plugin", so, gen.data)
    if _no_gpu():
        assert r[0] == H.Atom("error") and "no HIP device" in r[1]
    else:
        assert r[0] == H.Atom("ok")
        assert _lib  # (the GPU suite goes on from here: tests/test_gpu_nif_shim.py)


def test_without_a_gpu_the_shims_fail_loudly(mods):
    if not _no_gpu():
        pytest.skip("a GPU is visible; the GPU suite exercises the calls")
    r = mods["HipNative"].call("model_create", 2, np.arange(16.0))
    assert r[0] == H.Atom("error") and "no HIP device" in r[1] and "no CPU fallback" in r[1]
    q = np.zeros(3)
    with pytest.raises(H.Raised) as ei:
        mods["NativeTree"].call("init_trajectory_bin", q, q, q, 0.0)
    tag, code, msg = ei.value.reason
    assert tag == H.Atom("exmc_hip_error") and code == 2 and "no HIP device" in msg
    with pytest.raises(H.Raised):
        mods["NativeTree"].call("build_full_tree_bin", q, q, q, 0.0, q, q, np.zeros(1), q, q, q,
                                np.zeros(1), q, np.ones(3), 0.0, 3, 3, 42)
