"""Term order of node maps above 32 keys (VERDICT r4 "missing" 7). compiler.ex:176-180 builds the
terms in Map.values order and sum_logps (compiler.ex:394-395) adds them in that order; up to 32 keys
an Erlang map is sorted, above it iterates in the order of its internal hash, which only the VM knows.
HipExport.to_json writes Map.keys(nodes) down as "term_order" and the generator follows it; an IR
built without a VM keeps the sorted-id order (the documented fallback)."""
import json

import numpy as np
import pytest

import gen_checker as GC
import oracle as O
from exmc_amd import codegen as cg

DET = O.Cfg(1, 1)
N_OBS = 17


def _ir():
    """6 free Normal rvs + 17 observed Normal rvs with their obs nodes = 40 nodes, d = 6 (one lane per
    chain). Unit scales: every term is -0.5 * (z^2 + LOG_2PI_F32) with z a plain difference, so a
    numpy restatement gives the same bits term by term and only the ORDER of the final sum is open."""
    ir = cg.IR()
    for j in range(6):
        ir.rv("m_%d" % j, "normal", {"mu": 0.0, "sigma": 1.0})
    rng = np.random.default_rng(3)
    vals = rng.normal(size=N_OBS) * np.logspace(-3, 3, N_OBS)     # magnitudes spread: the order shows in the last bits
    for k in range(N_OBS):
        ir.rv("y_%02d" % k, "normal", {"mu": "m_%d" % (k % 6), "sigma": 1.0})
        ir.obs("obs_%02d" % k, "y_%02d" % k, float(vals[k]))
    return ir, vals


def _fold(order, q, vals):
    """sum_logps: Enum.reduce(list, fn x, acc -> Nx.add(x, acc)) over the terms in `order`."""
    t = {}
    for j in range(6):
        t["m_%d" % j] = -0.5 * (q[j] * q[j] + cg.LOG_2PI_F32)
    for k in range(N_OBS):
        z = vals[k] - q[k % 6]
        t["obs_%02d" % k] = -0.5 * (z * z + cg.LOG_2PI_F32)
    ids = [i for i in order if i in t]       # observed rvs and det nodes have no term of their own
    acc = t[ids[0]]
    for i in ids[1:]:
        acc = t[i] + acc
    return acc


def test_terms_follow_the_exported_map_order():
    ir, vals = _ir()
    ids = sorted(ir.nodes)
    assert len(ids) == 40 > cg.MAX_NODES_SORTED
    rng = np.random.default_rng(8)
    orders = [ids, list(reversed(ids))] + [list(rng.permutation(ids)) for _ in range(2)]
    qs = [rng.normal(size=6) * 2.0 for _ in range(40)]
    seen = []
    digests = set()
    for order in orders:
        ir2, _ = _ir()
        ir2.order(order)
        gen = cg.generate(ir2, ncp=False)
        digests.add(gen.digest)
        m = GC.model(gen, 1)
        lps = []
        for q in qs:
            lp, g = m.logp_grad(q, DET)
            assert lp == _fold(order, q, vals)          # bit for bit: the sum runs in the exported order
            lps.append(lp)
        seen.append(lps)
    assert len(digests) == len(orders)                  # the order is part of the generated text
    # the test has power: some other order gives other bits at some point
    assert any(a != b for lps in seen[1:] for a, b in zip(seen[0], lps))
    assert np.allclose(seen[0], seen[1], rtol=1e-14)
    # no exported order = sorted ids (the fallback an IR built without a VM takes)
    assert cg.generate(_ir()[0], ncp=False).digest == cg.generate(_ir()[0].order(ids), ncp=False).digest


def test_term_order_travels_in_the_json_document_and_is_checked():
    ir, _ = _ir()
    ids = sorted(ir.nodes)
    order = list(np.random.default_rng(1).permutation(ids))
    order = [str(i) for i in order]
    nodes = {}                                          # the document HipExport.to_json writes
    for id_, n in ir.nodes.items():
        if n["op"] == "rv":
            nodes[id_] = {"op": "rv", "dist": n["dist"], "transform": None, "params": n["params"]}
        else:
            nodes[id_] = {"op": "obs", "target": n["target"], "value": float(n["value"])}
    doc = {"ncp": False, "term_order": order, "nodes": nodes}
    ir2 = cg.ir_from_json(json.loads(json.dumps(doc)))
    assert ir2.term_order == order
    ir3, _ = _ir()
    assert cg.generate(ir2, ncp=False).digest == cg.generate(ir3.order(order), ncp=False).digest
    with pytest.raises(cg.CodegenError):
        cg.IR.order(_ir()[0], ids[:-1])                 # not every id
    with pytest.raises(cg.CodegenError):
        cg.IR.order(_ir()[0], ids[:-1] + [ids[0]])      # a duplicate
    # a map of <= 32 keys IS sorted on the BEAM: an exporter that claims otherwise is wrong
    small = cg.IR()
    small.rv("a", "normal", {"mu": 0.0, "sigma": 1.0})
    small.rv("b", "normal", {"mu": 0.0, "sigma": 1.0})
    small.order(["b", "a"])
    with pytest.raises(cg.CodegenError):
        cg.generate(small, ncp=False)
    small.order(["a", "b"])
    assert cg.generate(small, ncp=False).d == 2


def test_exporter_writes_the_order():
    src = open(GC.ROOT + "/elixir/lib/exmc/nuts/hip_export.ex").read()
    assert '"term_order" => Map.keys(nodes)' in src
