"""The Elixir side of the boundary as source files (elixir/): what can be checked
without a BEAM. Every `def name(args), do: :erlang.nif_error(:nif_not_loaded)` stub of
elixir/lib/exmc/nuts/hip_native.ex against the ErlNifFunc table of c_src/exmc_hip_nif.c (parsed from
the C source AND read from the compiled shim's nif_init()), every HipNative call in hip_sampler.ex
against the stubs' arities, balanced blocks, and the patch files' hunk headers. The reference tree is
not read."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "elixir")
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _read(*parts):
    with open(os.path.join(*parts)) as f:
        return f.read()


def _strip(src):
    """Elixir source without comments, heredocs, strings and charlists (good enough for counting)."""
    src = re.sub(r'"""[\s\S]*?"""', '""', src)
    src = re.sub(r'~c"[^"]*"', '""', src)
    src = re.sub(r'"(?:\\.|[^"\\])*"', '""', src)
    return "\n".join(ln.split("#", 1)[0] for ln in src.splitlines())


def _split_args(argstr):
    out, depth, cur = [], 0, ""
    for ch in argstr:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def _stubs():
    src = _strip(_read(EX, "lib", "exmc", "nuts", "hip_native.ex"))
    stubs = {}
    for m in re.finditer(r"def\s+([a-z_]+)\(([^)]*)\)\s*,?\s*do:\s*:erlang\.nif_error\(:nif_not_loaded\)", src):
        stubs[m.group(1)] = len(_split_args(m.group(2)))
    return stubs


def _c_table():
    src = _read(ROOT, "c_src", "exmc_hip_nif.c")
    body = src[src.index("static ErlNifFunc nif_funcs[]"):]
    body = body[:body.index("};")]
    return {m.group(1): int(m.group(2)) for m in re.finditer(r'\{"([a-z_]+)",\s*(\d+),\s*([a-z_]+),', body)}


def test_hip_native_stubs_equal_the_nif_table():
    stubs, table = _stubs(), _c_table()
    assert len(table) >= 16
    assert stubs == table
    src = _read(EX, "lib", "exmc", "nuts", "hip_native.ex")
    assert "defmodule Exmc.NUTS.HipNative do" in src and "@on_load :load_nif" in src
    assert ":erlang.load_nif" in src
    # the module name the shim registers is this module
    assert "ERL_NIF_INIT(Elixir.Exmc.NUTS.HipNative," in _read(ROOT, "c_src", "exmc_hip_nif.c")


from test_nif_shim import mods  # noqa: E402,F401  (the fixture that compiles both shims)


def test_stubs_equal_the_compiled_shims_entry(mods):  # noqa: F811
    """the same comparison against what nif_init() of the COMPILED shim returns (tests/test_nif_shim.py
    builds it against the declaration header)"""
    assert {n: a for n, a, _ in mods["HipNative"].table()} == _stubs()


@pytest.mark.parametrize("name", ["hip_native.ex", "hip_export.ex", "hip_sampler.ex"])
def test_blocks_balance(name):
    src = _strip(_read(EX, "lib", "exmc", "nuts", name))
    # `do` that opens a block (not the keyword form `do:`) against `end`; fn ... end counted too
    opens = len(re.findall(r"\bdo\b(?!:)", src)) + len(re.findall(r"\bfn\b", src))
    ends = len(re.findall(r"\bend\b", src))
    assert opens == ends, (name, opens, ends)
    for a, b in ("()", "[]", "{}"):
        assert src.count(a) == src.count(b), (name, a)
    assert src.lstrip().startswith("defmodule Exmc.NUTS.")


def test_sampler_calls_the_stubs_with_their_arities():
    stubs = _stubs()
    src = _strip(_read(EX, "lib", "exmc", "nuts", "hip_sampler.ex"))
    calls = []
    for m in re.finditer(r"HipNative\.([a-z_]+)\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        calls.append((m.group(1), len(_split_args(src[m.end():i - 1]))))
    assert {n for n, _ in calls} >= {"warmup", "sample_chains", "sample_independent", "stream_begin", "stream_run",
                                     "model_create", "model_create_plugin", "model_set_flat_order"}
    for n, a in calls:
        assert stubs.get(n) == a, (n, a, stubs.get(n))


def test_exporter_names_only_distributions_the_generator_covers():
    from exmc_amd import codegen
    src = _read(EX, "lib", "exmc", "nuts", "hip_export.ex")
    names = re.findall(r'Exmc\.Dist\.[A-Za-z0-9]+ => "([a-z0-9_]+)"', src)
    assert len(names) == len(set(names)) >= 19
    gen = _read(ROOT, "exmc_amd", "codegen.py")
    for n in names:
        assert ('"%s"' % n) in gen, n
    assert codegen.VECTOR_DISTS == ("gaussian_random_walk", "mv_normal", "dirichlet")
    for key in ('"ncp"', '"data"', '"nodes"', '"op"', '"dist"', '"transform"', '"params"', '"target"', '"value"',
                '"info"', '"fun"', '"args"', '"f32"'):
        assert key in src, key          # the keys codegen.ir_from_json reads


@pytest.mark.parametrize("name,target", [("sampler.ex.diff", "lib/exmc/nuts/sampler.ex"),
                                         ("compiler.ex.diff", "lib/exmc/compiler.ex")])
def test_patches_are_well_formed_unified_diffs(name, target):
    src = _read(EX, "patches", name)
    assert "--- a/%s" % target in src and "+++ b/%s" % target in src
    hunks = list(re.finditer(r"^@@ -(\d+),(\d+) \+(\d+),(\d+) @@$", src, re.M))
    assert hunks
    body_lines = [ln for ln in src.splitlines() if ln and not ln.startswith(("#", "---", "+++", "@@"))]
    assert all(ln[0] in "+- " for ln in body_lines), [ln for ln in body_lines if ln[0] not in "+- "][:3]
    added = "\n".join(ln[1:] for ln in body_lines if ln[0] == "+")
    assert "Exmc.NUTS.HipSampler." in added
    # context / removed lines quoted from the reference stay a handful (the patch is an anchor, not a copy)
    assert sum(1 for ln in body_lines if ln[0] in "- ") <= 8
    # the line counts of every hunk header are exact and the new-file positions add up (round 6: `patch`
    # refuses a hunk whose counts are off; the patches were dry-run against the reference's two files)
    lines = src.split("\n")
    at = [i for i, ln in enumerate(lines) if ln.startswith("@@ ")]
    offset = 0
    last = 0
    for n, i in enumerate(at):
        end = at[n + 1] if n + 1 < len(at) else len(lines)
        body = [ln for ln in lines[i + 1:end] if ln]
        o0, oc, n0, nc = map(int, re.match(r"^@@ -(\d+),(\d+) \+(\d+),(\d+) @@$", lines[i]).groups())
        assert oc == sum(1 for ln in body if ln[0] in " -") and nc == sum(1 for ln in body if ln[0] in " +"), lines[i]
        assert n0 == o0 + offset and o0 > last, lines[i]
        offset += nc - oc
        last = o0


def test_tree_patch_adds_the_fused_chain_clause():
    """elixir/patches/tree.ex.diff: one hunk, between the Nx.Vulkan clause of do_dispatch/10 and the fall-through
    (tree.ex:653-656); it calls the stub with its arity and returns the tuple both existing clauses return."""
    src = _read(EX, "patches", "tree.ex.diff")
    assert "--- a/lib/exmc/nuts/tree.ex" in src and "+++ b/lib/exmc/nuts/tree.ex" in src
    hunks = list(re.finditer(r"^@@ -(\d+),(\d+) \+(\d+),(\d+) @@$", src, re.M))
    assert len(hunks) == 1
    o0, oc, n0, nc = map(int, hunks[0].groups())
    body = [ln for ln in src[hunks[0].end():].split("\n") if ln]
    assert all(ln[0] in "+ " for ln in body)
    assert oc == sum(1 for ln in body if ln[0] == " ") == 4 and nc == len(body) and o0 == n0 == 653
    added = _strip("\n".join(ln[1:] for ln in body if ln[0] == "+"))
    m = re.search(r"HipNative\.leapfrog_chain_normal\(", added)
    i, depth = m.end(), 1
    while depth:
        depth += {"(": 1, ")": -1}.get(added[i], 0)
        i += 1
    assert len(_split_args(added[m.end():i - 1])) == _stubs()["leapfrog_chain_normal"] == 7
    assert "Exmc.NUTS.HipSampler.available?()" in added and "d <= 256" in added
    assert "ten.(q_chain, {k, d}), ten.(p_chain, {k, d}), ten.(logp_chain, {k}), ten.(grad_chain, {k, d})" in added
    opens = len(re.findall(r"\bdo\b(?!:)", added)) + len(re.findall(r"\bfn\b", added))
    assert opens == len(re.findall(r"\bend\b", added))
