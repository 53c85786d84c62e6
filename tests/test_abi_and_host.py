"""CPU tests: the C-ABI library loads and exports every symbol include/exmc_hip.h declares, the
product path fails loudly without a GPU (no CPU fallback), and the host-side mirror of
Exmc.NUTS.Sampler marshals what the reference API promises. No compute calls here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "exmc_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(exmc_hip_\w+)\s*\(", txt)))


def test_build_entry_compiles_for_gfx950():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    from exmc_amd import build as b
    assert os.path.exists(b.OUT)
    assert "--offload-arch=gfx950" in b.FLAGS and "-ffp-contract=off" in b.FLAGS
    # the code object inside the .so targets gfx950
    out = subprocess.run(["strings", "-n", "6", b.OUT], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_library_exports_every_declared_symbol():
    from exmc_amd import _lib
    L = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), s
    assert sorted(_lib.EXPORTS) == syms


def test_header_is_plain_c():
    """The boundary is a C ABI: the header must compile as C with no torch/HIP types."""
    src = '#include "exmc_hip.h"\nint main(void){return (int)sizeof(exmc_hip_opts) == 0;}\n'
    p = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-x", "c", "-", "-o", "/dev/null"], input=src, text=True, capture_output=True)
    assert p.returncode == 0, p.stderr


def test_struct_layouts_match_header():
    from exmc_amd import _lib
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "exmc_hip.h"\n'
           'int main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(exmc_hip_opts),'
           'offsetof(exmc_hip_opts, seed), sizeof(exmc_hip_tuning),'
           'offsetof(exmc_hip_tuning, warmup_divergences), sizeof(exmc_hip_trace),'
           'offsetof(exmc_hip_trace, energy));return 0;}\n')
    exe = "/tmp/exmc_layout_check"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", exe],
                   input=src.encode(), check=True)
    got = [int(x) for x in subprocess.check_output([exe]).split()]
    assert got == [C.sizeof(_lib.Opts), _lib.Opts.seed.offset, C.sizeof(_lib.Tuning),
                   _lib.Tuning.warmup_divergences.offset, C.sizeof(_lib.Trace),
                   _lib.Trace.energy.offset]


def test_fails_loudly_without_a_gpu():
    """No HIP device => model creation returns EXMC_ERR_NO_DEVICE and the Python mirror raises;
    nothing silently falls back to the CPU."""
    from exmc_amd import _lib, models, sampler
    L = _lib.load()
    if L.exmc_hip_device_count() > 0:
        pytest.skip("a GPU is present; covered by the -m gpu tests")
    h = C.c_void_p()
    spec = models.eight_schools()
    rc = L.exmc_hip_model_create(spec.kind, spec.d, spec.data.ctypes.data_as(C.POINTER(C.c_double)),
                                 16, 0, C.byref(h))
    assert rc == _lib.ERR_NO_DEVICE and not h.value
    assert b"no CPU fallback" in L.exmc_hip_last_error()
    with pytest.raises(_lib.ExmcHipError):
        sampler.sample(spec, spec.default_init, dict(num_warmup=10, num_samples=10))
    with pytest.raises(_lib.ExmcHipError):
        sampler.sample_chains(spec, 4, dict(num_warmup=10, num_samples=10))
    assert L.exmc_hip_model_dim(None) == -1


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under exmc_amd/ or include/ may reference it."""
    bad = []
    for base in ("exmc_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"exmc_oracle|libexmc_oracle|import oracle|from oracle|exo_", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_model_specs_and_point_map():
    from exmc_amd import models
    es = models.eight_schools()
    assert es.d == 10 and es.var_names[:2] == ["mu", "tau"]
    # point_map.ex:37: flat order = ids sorted as strings; eight_schools is already sorted
    assert es.flat_order() == list(range(10))
    q = es.to_unconstrained(es.default_init)
    assert np.array_equal(q, np.zeros(10))          # log(tau = 1) = 0
    x = es.constrain(np.array([[0.5, np.log(3.0)] + [0.0] * 8]))
    assert abs(x[0, 1] - 3.0) < 1e-12 and x[0, 0] == 0.5
    sv = models.sv(np.linspace(-1, 1, 100))
    names = [sv.var_names[i] for i in sv.flat_order()]
    assert names[:5] == ["nu", "s_1", "s_10", "s_100", "s_11"] and names[-1] == "sigma"
    with pytest.raises(KeyError):
        es.to_unconstrained({"mu": 0.0})            # Map.fetch! semantics
    with pytest.raises(ValueError):
        models.sv([0.0] * 99)


def test_sampler_option_defaults_match_reference():
    from exmc_amd import sampler
    o = sampler._merge_opts({"seed": 7})
    assert (o["num_warmup"], o["num_samples"], o["max_tree_depth"], o["target_accept"], o["seed"]) \
        == (1000, 1000, 10, 0.8, 7)       # sampler.ex:16-23
    c = sampler._c_opts(o)
    assert (c.num_warmup, c.num_samples, c.max_tree_depth, c.seed, c.lanes_per_chain) == (1000, 1000, 10, 7, 0)
    with pytest.raises(ValueError):
        sampler._tuning_struct({"epsilon": 0.1, "inv_mass": np.ones((2, 3))}, 3)
    # a dense tuning (d x d covariance) contributes its diagonal to the struct (sampler.ex:236-240)
    t = sampler._tuning_struct({"epsilon": 0.1, "inv_mass": np.diag([1.0, 2.0, 3.0]) + 0.1}, 3)
    assert [t.inv_mass[i] for i in range(3)] == [1.1, 2.1, 3.1]


def test_plugin_carries_its_kernels_as_code_objects_not_host_stubs():
    """A generated model's plug-in (exmc_amd/codegen.py build_plugin, round 4): the model-dependent
    kernels are compiled device-only, embedded as code objects behind `exmc_blob_table` and launched by
    name through hipModuleLaunchKernel -- so the library has the table, gfx950 code objects behind it,
    every C-ABI entry point, and NO host stub of a model-dependent kernel (the model-independent
    kernels of exmc_common.o keep theirs)."""
    import subprocess
    from exmc_amd import _lib, codegen
    so = codegen.build_plugin(codegen.generate(codegen.simple_ir()))
    syms = subprocess.run(["nm", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    assert " exmc_blob_table" in syms
    stubs = [ln for ln in syms.splitlines() if "__device_stub__" in ln]
    assert stubs and all(any(k in ln for k in ("ess_", "rhat_kernel", "rank_scores", "full_tree", "subtree", "traj_build", "chain_normal"))
                         for ln in stubs), stubs
    for k in ("nuts_kernel", "warmup_kernel", "multi_step_kernel", "logp_grad_kernel", "init_chains_kernel", "find_eps_kernel"):
        assert not any(k in ln for ln in stubs), k
    blob = open(so, "rb").read()
    assert blob.count(b"amdgcn-amd-amdhsa--gfx950") >= 5          # one code object per part (+ the host unit's own)
    for k in (b"nuts_kernel", b"warmup_kernel", b"find_eps_kernel"):
        assert k in blob                                          # the kernels' device-side names, looked up at launch
    P = _lib.bind(so)
    for name in _lib.EXPORTS:
        getattr(P, name)


def test_plugin_build_form_is_part_of_the_cache_tag_and_missing_kernels_fail_the_build(tmp_path, monkeypatch):
    """(a) the library of one build form (code objects / host stubs / one unit) must never be
    handed out for another -- the form is in the cache tag; (b) a kernel the host unit can launch but
    no part defines is a BUILD error, as the link error it used to be, not a launch-time one."""
    import re
    import subprocess
    from exmc_amd import build as _build
    from exmc_amd import codegen
    gen = codegen.generate(codegen.simple_ir())
    monkeypatch.delenv("EXMC_PLUGIN_STUBS", raising=False)
    monkeypatch.delenv("EXMC_PLUGIN_ONE_TU", raising=False)
    d_mod = codegen.plugin_paths(gen)[0]
    monkeypatch.setenv("EXMC_PLUGIN_STUBS", "1")
    d_stub = codegen.plugin_paths(gen)[0]
    monkeypatch.setenv("EXMC_PLUGIN_ONE_TU", "1")
    d_one = codegen.plugin_paths(gen)[0]
    assert len({d_mod, d_stub, d_one}) == 3
    monkeypatch.delenv("EXMC_PLUGIN_STUBS")
    monkeypatch.delenv("EXMC_PLUGIN_ONE_TU")
    # (b) a host object that names a kernel nobody defines
    src = tmp_path / "h.c"
    src.write_text('const char* a = "_ZN4exmc11nuts_kernelINS_6CustomILi1EEELi1ELi2ELb0ELb0EEEvNS_10NutsParamsENT_6ConstsE";\n'
                   'const char* b = "_ZN4exmc13bogus_kernel_that_no_part_definesEv";\n')
    host = tmp_path / "h.o"
    subprocess.check_call(["gcc", "-c", "-o", str(host), str(src)])
    part = tmp_path / "p.o"
    psrc = tmp_path / "p.c"
    psrc.write_text("void _ZN4exmc11nuts_kernelINS_6CustomILi1EEELi1ELi2ELb0ELb0EEEvNS_10NutsParamsENT_6ConstsE(void) {}\n")
    subprocess.check_call(["gcc", "-c", "-o", str(part), str(psrc)])
    common = _build.build_common()
    with pytest.raises(RuntimeError, match="bogus_kernel_that_no_part_defines"):
        codegen._check_module_kernels(str(host), [str(part)], common)
    host2 = tmp_path / "h2.o"
    src.write_text(re.sub(r"const char\* b.*\n", "", src.read_text()))
    subprocess.check_call(["gcc", "-c", "-o", str(host2), str(src)])
    codegen._check_module_kernels(str(host2), [str(part)], common)     # every name defined: no error
