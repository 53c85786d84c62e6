"""Several dimensions per lane: the lane layout for generated models of any size (d > 16).

The one-lane layout of codegen.py holds the whole position in one lane's registers (d <= 20) and the
plate layout of codegen_vec.py gives every dimension a lane of a 16-lane row (d <= 16). The three
larger BASELINE models are hand-written lane layouts (exmc_models.hpp: SV<64>, Radon<64>,
Logistic<16>): a chain over G lanes, dimension i in slot i / G of lane i % G, the model's repeated
terms spread over the lanes. This module derives that layout from the expression graph of ANY
Builder IR (compiler.ex:176-269 walks the same nodes), so that stochastic volatility (a 100-step
random walk with StudentT observations), the radon model (85 county intercepts, 919 observations)
and the 500 x 20 logistic regression compile from node lists like every smaller model.

  * The log-density is a sum of terms (compiler.ex:394-395). Terms that are themselves sums -- a
    vector observation's elements, the steps of a GaussianRandomWalk, the reduction inside a Custom
    closure -- are split into their summands: the *units*.
  * Units whose expression DAGs are identical up to their leaves form a *family*. Inside a family a
    node that is the SAME graph node in every unit (the graph is hash-consed, so a shared
    hyper-parameter and everything computed from shared hyper-parameters alone is one node) is
    *uniform*: it is evaluated once per leapfrog, outside the family's loop, by every lane alike.
    The remaining nodes are the family's template; its leaves are uniform values, per-unit
    constants (columns of a table folded from the data when the model is generated) and free
    variables that differ from unit to unit (*gathered*: read through a per-unit index).
  * Unit u of a family runs on lane u % G in slot u / G of a counted loop. The position is
    published in an LDS strip of d doubles per chain, so a gathered leaf is one LDS read at a
    per-unit address and a shared variable a broadcast read.
  * Reverse mode per family (codegen._Grad on the template): the adjoints of the uniform inputs are
    summed per lane over its slots and cross the chain group in ONE butterfly together with the
    lane's partial log-density; the adjoints of gathered leaves go to a per-family LDS strip, one
    cell per unit, and the lane that owns a dimension adds the cells that belong to it in a fixed
    order (family, leaf position, unit) read from an index table padded to the widest lane.
  * The uniform part (hyper-priors, everything derived from shared variables, terms that occur
    once) is differentiated as in the one-lane layout, seeded with the reduced adjoints.

What keeps the generated kernels near the hand-written ones (measured there):

  * Families INSIDE the uniform part (_uniform_families): a sum of >= 4 like terms of shared values
    (the quotients of a Lanczos series) is evaluated one term per lane, value and partials reduced
    in a butterfly of their own; the tangent of the sum is the chain rule over its inputs.
  * Batches (ustmts): the chain-scalar exp / log / reciprocals of one dependency level are
    evaluated together, argument i on lane i, and broadcast back.
  * Quotients (build_once.quotient): one reciprocal per distinct denominator serves value and
    adjoint; by a constant it is folded into the tables, by a uniform value it leaves the loop
    (emit_family hoists what does not change from unit to unit).
  * Zero factors (_split_by_zero_factors, build_once.make / bounds): units that differ in WHICH of
    their constant factors are zero form separate families; 0 * x is folded when interval bounds
    prove x finite (each half of a Bernoulli likelihood evaluates one logarithm).
  * Fetching ahead (emit_family): short table rows and gathered variables of a block of slots are
    loaded before the block's arithmetic (a lone wave per SIMD cannot hide the round trips).
  * One chain on a whole wavefront (EXMC_GEN_G0 / _NG / _XGROUP): for the shared warmup a layout of
    fewer than 64 lanes per chain spreads the slots of every family over the 64 / G lane groups and
    adds the groups' sums in group order.
  * waves_per_simd = 2 (generate): the register cap of two resident waves, with the plug-in's LDS
    sized so that eight workgroups fit a CU (exmc_models.hpp EXMC_GEN_LDS_*).

Numeric contract: the lane contract of DESIGN.md section 2 -- per lane left to right over its slots,
then the xor butterfly over the group -- so the sum of a family's terms is NOT the reference's
left-to-right Nx.sum (nor, for models above 32 nodes, the hash order of an Erlang map, which is not
restated: terms are taken in sorted-id order); the difference is rounding of a sum (a few ulp of
the log-density). The host checker (tests/gen_checker.py) runs the same text over G virtual lanes
in that order and the GPU equals it bit for bit; against the hand-written models of the oracle the
generated log-density and gradient agree to 1e-12 relative.
"""
import math

import numpy as np

from . import codegen as cg

MIN_FAMILY = 4          # fewer units than this are evaluated by every lane (the uniform part)
# doubles of table rows / gathered variables fetched ahead per lane (emit_family), by resident waves
# per SIMD: a lone wave has 512 registers and nobody to hide its latency, a pair 256 each
PAIR_MIN_COLS = 8      # families of at least this many per-unit columns store them in interleaved pairs (generate)
PREFETCH_DOUBLES = {1: int(__import__("os").environ.get("EXMC_GEN_PREFETCH", "64")), 2: 24}


class _LGraph(cg._Graph):
    """_Graph with the template's leaves: `ext j` (uniform dynamic input j), `uc k` (uniform
    constant k), `col c` (column c of the family's table), `gat p` (gathered variable p) and,
    in the uniform graph, `red j` (reduced sum j after the butterfly)."""

    LEAF_CONST = {"lit": True, "data": True, "q": False, "ext": False, "uc": True, "col": True,
                  "gat": False, "red": False, "wred": False}

    def _node(self, op, *args):
        k = (op,) + args
        i = self.key.get(k)
        if i is None:
            i = len(self.ops)
            self.ops.append(k)
            c = self.LEAF_CONST[op] if op in self.LEAF_CONST else all(self.const[a] for a in args)
            self.const.append(c)
            self.key[k] = i
        return i


_LEAVES = ("lit", "data", "q", "ext", "uc", "col", "gat", "red", "wred")


def _np_eval(g, nodes, leaf):
    """Values of the const nodes `nodes` (and what they need) with numpy float64 semantics;
    leaf(op) gives the value of a leaf (a scalar or an array over units). exp / log of a constant
    of the data come from numpy here (a rounding-level difference in a constant, like the
    Cholesky of codegen's MvNormal)."""
    memo = {}
    order, seen, stack = [], set(), list(nodes)
    while stack:
        n = stack.pop()
        if n in seen:
            continue
        op = g.ops[n]
        if op[0] in _LEAVES:
            seen.add(n)
            memo[n] = np.float64(float.fromhex(op[1])) if op[0] == "lit" else leaf(op)
            continue
        pend = [a for a in op[1:] if a not in seen]
        if pend:
            stack.append(n)
            stack.extend(pend)
        else:
            seen.add(n)
            order.append(n)
    f64 = lambda x: np.asarray(x, dtype=np.float64)   # noqa: E731
    with np.errstate(all="ignore"):
        for n in order:
            op = g.ops[n]
            a = [f64(memo[x]) for x in op[1:]]
            k = op[0]
            if k == "add": v = a[0] + a[1]
            elif k == "sub": v = a[0] - a[1]
            elif k == "mul": v = a[0] * a[1]
            elif k == "div": v = a[0] / a[1]
            elif k == "neg": v = -a[0]
            elif k == "exp": v = np.exp(a[0])
            elif k == "log": v = np.log(a[0])
            elif k == "log1p": v = np.log1p(a[0])
            elif k == "erf":
                from scipy.special import erf
                v = erf(a[0])
            elif k == "abs": v = np.abs(a[0])
            elif k == "max": v = np.fmax(a[0], a[1])
            elif k == "min": v = np.fmin(a[0], a[1])
            elif k == "sel_gt": v = np.where(a[0] > a[1], a[2], a[3])
            else:
                raise cg.CodegenError("cannot fold %s" % k)
            memo[n] = v
    return memo


def _flatten(g, root, full):
    """The summands of a term: through registered sums (codegen._sum_left) and through `add` nodes
    that lead to one; the result of a Custom closure that holds no registered sum is a reduction
    written by hand (Enum.reduce with Nx.add, validate_posteriordb.exs:279-295) and is split at
    every `add` at its top."""
    worthy = {}

    def is_worthy(n):
        stack = [n]
        while stack:
            x = stack[-1]
            if x in worthy:
                stack.pop()
                continue
            if x in g.sums:
                worthy[x] = True
                stack.pop()
            elif g.ops[x][0] == "add":
                a, b = g.ops[x][1:]
                if a in worthy and b in worthy:
                    worthy[x] = worthy[a] or worthy[b]
                    stack.pop()
                else:
                    stack.extend(c for c in (a, b) if c not in worthy)
            else:
                worthy[x] = False
                stack.pop()
        return worthy[n]

    full = full and not is_worthy(root)
    out, stack = [], [root]
    while stack:
        n = stack.pop()
        if n in g.sums:
            stack.extend(reversed(g.sums[n]))
        elif g.ops[n][0] == "add" and (full or is_worthy(n)):
            stack.extend(reversed(g.ops[n][1:]))
        else:
            out.append(n)
    return out


def _signature(g, root, wild=False):
    """(shape, ids): the unit's DAG in post-order with local numbering; shape is what two units of
    a family share, ids[k] the graph node behind local index k. wild: a literal is a per-unit
    constant like a datum (the coefficients of a series written out term by term)."""
    local, shape, ids = {}, [], []
    stack = [(root, False)]
    is_wild = (lambda a: g.ops[a][0] == "lit") if wild else (lambda a: False)
    while stack:
        n, done = stack.pop()
        if n in local:
            continue
        op = g.ops[n]
        if op[0] in ("lit", "q", "data"):
            local[n] = len(shape)
            shape.append((("wlit",) if wild else ("lit", op[1])) if op[0] == "lit" else (op[0],))
            ids.append(n)
        elif not done:
            stack.append((n, True))
            stack.extend((a, False) for a in reversed(op[1:]) if a not in local and not is_wild(a))
        else:
            args = []
            for a in op[1:]:
                if is_wild(a):            # every occurrence of a literal is a leaf of its own: the graph
                    args.append(len(shape))   # shares equal literals, the terms of a series do not
                    shape.append(("wlit",))
                    ids.append(a)
                else:
                    args.append(local[a])
            local[n] = len(shape)
            shape.append((op[0],) + tuple(args))
            ids.append(n)
    return tuple(shape), ids


class _Family:
    pass


def _nonzero_const(g, n):
    """A constant node whose value is finite and not zero (so that its reciprocal is a constant too)."""
    v = float(_np_eval(g, [n], lambda op: np.float64(g.data[op[1]]))[n])
    return math.isfinite(v) and v != 0.0 and math.isfinite(1.0 / v)


def _forward_tangents(g, roots, spread=None):
    """{node: {shared variable index: tangent node}} for every non-constant node the roots need.
    Local rules mirror codegen._Grad (max / min pass the tangent on strict inequality only).
    spread: {sum node: [(input node, node holding d sum / d input)]} for the sums evaluated over the
    lanes (_uniform_families): their tangent is the chain rule over their inputs."""
    spread = spread or {}
    need, stack = set(), list(roots)
    while stack:
        n = stack.pop()
        if n in need or g.const[n]:
            continue
        need.add(n)
        if n in spread:
            stack.extend(e for e, _ in spread[n])
        elif g.ops[n][0] not in _LEAVES:
            stack.extend(g.ops[n][1:])
    zero = g.lit(0.0)
    tan = {}

    def both(ta, tb, fa, fb, fab):
        out = {}
        for v in set(ta) | set(tb):
            if v in ta and v in tb:
                out[v] = fab(ta[v], tb[v])
            elif v in ta:
                out[v] = fa(ta[v])
            else:
                out[v] = fb(tb[v])
        return out
    for n in sorted(need):
        op = g.ops[n]
        k = op[0]
        if k == "q":
            tan[n] = {op[1]: g.lit(1.0)}
            continue
        if k in _LEAVES:
            tan[n] = {}
            continue
        if n in spread:
            r = {}
            for e, part in spread[n]:
                for v, x in tan.get(e, {}).items():
                    term = g.mul(part, x)
                    r[v] = term if v not in r else g.add(r[v], term)
            tan[n] = r
            continue
        a = op[1:]
        t = [tan.get(x, {}) for x in a]
        ident = lambda x: x   # noqa: E731
        if k == "add":
            r = both(t[0], t[1], ident, ident, g.add)
        elif k == "sub":
            r = both(t[0], t[1], ident, g.neg, g.sub)
        elif k == "neg":
            r = {v: g.neg(x) for v, x in t[0].items()}
        elif k == "mul":
            r = both(t[0], t[1], lambda x: g.mul(x, a[1]), lambda y: g.mul(a[0], y),
                     lambda x, y: g.add(g.mul(x, a[1]), g.mul(a[0], y)))
        elif k == "div":
            if not t[0] and g.const[a[0]] and _nonzero_const(g, a[0]):
                # c / b with a constant numerator: d/db = -y^2 / c, and 1 / c is folded with the data --
                # no second division at run time (a Lanczos series is eight such terms)
                rc = g.recip(a[0])
                r = {v: g.neg(g.mul(g.mul(g.mul(n, n), rc), y)) for v, y in t[1].items()}
            else:
                rb = g.recip(a[1])
                r = both(t[0], t[1], lambda x: g.mul(x, rb), lambda y: g.neg(g.mul(g.mul(n, y), rb)),
                         lambda x, y: g.mul(g.sub(x, g.mul(n, y)), rb))
        elif k == "exp":
            r = {v: g.mul(n, x) for v, x in t[0].items()}
        elif k == "log":
            ra = g.recip(a[0])
            r = {v: g.mul(x, ra) for v, x in t[0].items()}
        elif k == "log1p":
            ra = g.recip(g.add(g.lit(1.0), a[0]))
            r = {v: g.mul(x, ra) for v, x in t[0].items()}
        elif k == "erf":
            d = g.mul(g.lit(2.0 / math.sqrt(math.pi)), g.exp(g.neg(g.mul(a[0], a[0]))))
            r = {v: g.mul(x, d) for v, x in t[0].items()}
        elif k == "abs":
            r = {v: g.sel_gt(a[0], zero, x, g.sel_gt(zero, a[0], g.neg(x), zero)) for v, x in t[0].items()}
        elif k in ("max", "min"):
            first, second = (a[0], a[1]) if k == "max" else (a[1], a[0])
            r = both(t[0], t[1], lambda x: g.sel_gt(first, second, x, zero),
                     lambda y: g.sel_gt(second, first, y, zero),
                     lambda x, y: g.add(g.sel_gt(first, second, x, zero), g.sel_gt(second, first, y, zero)))
        elif k == "sel_gt":
            r = both(t[2], t[3], lambda x: g.sel_gt(a[0], a[1], x, zero), lambda y: g.sel_gt(a[0], a[1], zero, y),
                     lambda x, y: g.sel_gt(a[0], a[1], x, y))
        else:
            raise cg.CodegenError("no tangent rule for %s" % k)
        tan[n] = r
    return tan


def _uniform_families(g, roots):
    """Sums of >= MIN_FAMILY like terms in the non-constant graph under `roots` whose terms read
    shared values and constants only (no variable that differs from term to term); outermost first,
    and none whose inputs depend on another one's result."""
    reach, stack = set(), [r for r in roots]
    while stack:
        n = stack.pop()
        if n in reach or g.const[n]:
            continue
        reach.add(n)
        if g.ops[n][0] not in _LEAVES:
            stack.extend(g.ops[n][1:])
    found, absorbed = [], set()
    for V in sorted((n for n in reach if n in g.sums), reverse=True):
        if V in absorbed:
            continue
        terms = g.sums[V]
        dyn = [t for t in terms if not g.const[t]]
        if len(dyn) < MIN_FAMILY or len(set(dyn)) != len(dyn):
            continue
        sigs = [_signature(g, t, wild=True) for t in dyn]
        shape = sigs[0][0]
        if any(sg[0] != shape for sg in sigs[1:]):
            continue
        ids = [sg[1] for sg in sigs]
        uniform = [all(i[k] == ids[0][k] for i in ids) for k in range(len(shape))]
        if uniform[-1]:
            continue
        # the template: from the root down to uniform nodes; a variable of its own per term -> not this kind
        need, stack, ok = set(), [len(shape) - 1], True
        while stack:
            k = stack.pop()
            if k in need:
                continue
            need.add(k)
            if uniform[k]:
                continue
            if shape[k][0] == "q":
                ok = False
                break
            if shape[k][0] not in ("wlit", "data"):
                stack.extend(shape[k][1:])
        if not ok:
            continue
        f = _Family()
        f.shape, f.members, f.uniform, f.ids = shape, list(range(len(dyn))), uniform, ids
        f.out, f.const_part = V, [t for t in terms if g.const[t]]
        f.inputs = [ids[0][k] for k in sorted(need) if uniform[k] and not g.const[ids[0][k]]]
        found.append(f)
        # the partial sums between the terms and the terms' own nodes are no longer evaluated
        acc = terms[0]
        for t in terms[1:]:
            acc = g.key.get(("add", acc, t))
            if acc is not None:
                absorbed.add(acc)
        for i in ids:
            absorbed.update(i[k] for k in need if not uniform[k])
    # one level: drop a family whose inputs are computed from another family's sum
    outs = set(f.out for f in found)
    memo = {}

    def depends(n):
        stack = [n]
        while stack:
            x = stack[-1]
            if x in memo:
                stack.pop()
                continue
            if g.const[x] or g.ops[x][0] in _LEAVES:
                memo[x] = False
                stack.pop()
                continue
            if x in outs:
                memo[x] = True
                stack.pop()
                continue
            pend = [a for a in g.ops[x][1:] if a not in memo]
            if pend:
                stack.extend(pend)
            else:
                memo[x] = any(memo[a] for a in g.ops[x][1:])
                stack.pop()
        return memo[n]
    kept = []
    for f in sorted(found, key=lambda f: f.out):
        if any(depends(i) for i in f.inputs):
            outs.discard(f.out)
            memo.clear()
            continue
        kept.append(f)
    return kept


def _split_by_zero_factors(g, shape, members, sigs):
    """Units of one shape, split by WHICH of their constant factors are exactly zero (a 0 / 1 datum y
    in y * log p + (1 - y) * log(1 - p): the units with y = 1 and those with y = 0). In a subfamily
    such a factor is the same constant for every unit, and the template builder folds 0 * (a provably
    finite value) away -- each half of a Bernoulli likelihood then evaluates one logarithm, not two.
    Kept together when a part would have fewer than MIN_FAMILY units or there are more than four."""
    def costly(k, seen):          # a transcendental or a quotient under shape node k
        if k in seen:
            return False
        seen.add(k)
        e = shape[k]
        if e[0] in ("log", "exp", "log1p", "erf", "div"):
            return True
        return e[0] not in ("lit", "q", "data") and any(costly(a, seen) for a in e[1:])
    factors = []
    for k, e in enumerate(shape):
        if e[0] != "mul":
            continue
        for a, b in ((e[1], e[2]), (e[2], e[1])):
            node0 = sigs[members[0]][a]
            if (g.const[node0] and not g.const[sigs[members[0]][b]] and a not in factors
                    and costly(b, set())):         # (0 * a plain variable saves nothing worth a loop)
                factors.append(a)
    if not factors:
        return [members]
    leaf = lambda op: np.float64(g.data[op[1]])   # noqa: E731
    patterns = {}
    for m in members:
        nodes = [sigs[m][k] for k in factors]
        vals = _np_eval(g, nodes, leaf)
        patterns.setdefault(tuple(bool(vals[n] == 0.0) for n in nodes), []).append(m)
    if len(patterns) == 1 or len(patterns) > 4 or min(len(v) for v in patterns.values()) < MIN_FAMILY:
        return [members]
    return [patterns[p] for p in sorted(patterns, key=lambda p: patterns[p][0])]


def plan(g, term_roots, custom_roots, D, G):
    """Units, families and the uniform remainder of the graph `g` whose terms are `term_roots`."""
    units = []
    for t in term_roots:
        units.extend(_flatten(g, t, t in custom_roots))
    by_shape, sigs = {}, {}
    for pos, u in enumerate(units):
        if g.const[u]:
            continue
        shape, ids = _signature(g, u)
        sigs[pos] = ids
        by_shape.setdefault(shape, []).append(pos)
    families, in_family = [], set()
    for shape, all_members in by_shape.items():
        if len(all_members) < MIN_FAMILY:
            continue
        for members in _split_by_zero_factors(g, shape, all_members, sigs):
            first = sigs[members[0]]
            uniform = [all(sigs[m][k] == first[k] for m in members) for k in range(len(shape))]
            if uniform[-1]:
                continue            # the same node n times: n uniform terms
            f = _Family()
            f.shape, f.members, f.uniform = shape, members, uniform
            f.ids = [sigs[m] for m in members]
            families.append(f)
            in_family.update(members)
    families.sort(key=lambda f: f.members[0])
    scalar_units = [units[p] for p in range(len(units)) if p not in in_family]
    return families, scalar_units


def generate(g, term_roots, custom_roots, D, G, waves_per_simd=1):
    """-> dict(text, data, lanes, dpl, ...) for the lane layout. waves_per_simd: resident waves per
    SIMD the sampling kernel's register allocation must allow (ModelDefaults::kNutsWavesPerSimd):
    1 leaves the allocator the whole file; 2 caps it at 256 vector registers, which can pay when a
    launch has more wavefronts than the chip has SIMDs (chains x lanes / 64 > 1024) -- it does for
    the hand-written sv and logistic kinds, it does not for their generated forms, whose bodies
    then spill inside the leaf loop (profiles/r3_gen)."""
    if waves_per_simd not in (1, 2):
        raise cg.CodegenError("waves_per_simd must be 1 or 2")
    if G not in (16, 32, 64):
        raise cg.CodegenError("lanes per chain must be 16, 32 or 64")
    DPL = (D + G - 1) // G
    families, scalar_units = plan(g, term_roots, custom_roots, D, G)
    # ---- uniform constants: one table for the uniform part and every template ----
    uc_of, uc_vals = {}, []

    def uc_slot(key, value):
        if key not in uc_of:
            uc_of[key] = len(uc_vals)
            uc_vals.append(float(value))
        return uc_of[key]

    def g_const_value(n):
        return float(_np_eval(g, [n], lambda op: np.float64(g.data[op[1]]))[n])

    # ---- templates ----
    boundary = []            # distinct uniform dynamic nodes read by templates, in first-use order
    b_index = {}
    sh_off = D + 1           # LDS strip: [q (D)] [zero cell] [adjoint strips ...]
    dcols, icols = [], []    # table columns (each NPAD long), in emission order
    def build_template(f):
        """Twice: the second time the quotients by a denominator whose reciprocal the first pass
        needed anyway (the adjoint of log b, of another quotient) take that reciprocal too."""
        nonlocal sh_off
        build_once(f, frozenset())
        T = f.T
        shared = frozenset(k for k, node in f.tmap.items()
                           if T.key.get(("div", T.lit(1.0), node)) in f.live and not T.const[node])
        if shared:
            build_once(f, shared)
        # LDS strips of the gathered adjoints
        f.strip = []
        for p in range(len(f.gather)):
            f.strip.append(sh_off if f.gat_adj[p] is not None else -1)
            if f.gat_adj[p] is not None:
                sh_off += f.npad

    def build_once(f, share_recip):
        n, shape = len(f.members), f.shape
        S = (n + G - 1) // G
        f.n, f.S, f.npad = n, S, S * G
        root = len(shape) - 1
        # nodes of the template: reachable from the root without passing a uniform node
        need, stack = set(), [root]
        while stack:
            k = stack.pop()
            if k in need:
                continue
            need.add(k)
            if not f.uniform[k] and shape[k][0] not in ("lit", "wlit", "q", "data"):
                stack.extend(shape[k][1:])
        T = _LGraph()
        tmap, f.raw_cols, f.gather = {}, [], []     # raw data columns; gathered q indices per unit

        def const_leaf(op, raw=f.raw_cols):
            if op[0] == "col":
                return raw[-op[1] - 1]
            if op[0] == "uc":
                return np.float64(uc_vals[op[1]])
            raise cg.CodegenError("unexpected leaf %r in a constant" % (op,))

        def num_ok(node, T=T):
            """a constant of the template that is finite and non-zero in every unit, and so is 1 / it"""
            v = np.asarray(_np_eval(T, [node], const_leaf)[node], dtype=np.float64)
            with np.errstate(all="ignore"):
                return bool(np.all(np.isfinite(v)) and np.all(v != 0.0) and np.all(np.isfinite(1.0 / v)))

        def quotient(a, b, shared, T=T):
            """a / b of the template. One reciprocal per distinct denominator serves the value and the
            adjoint (the lane layout's own contract, like its fused multiply-adds: a * (1 / b) is within
            an ulp of a / b): by a constant the reciprocal is folded into the tables, by a uniform
            value it leaves the loop. c / b with a usable constant numerator stays a quotient -- its
            adjoint is -(y * y) / c and needs no reciprocal at all."""
            if T.const[b]:
                return T.mul(a, T.recip(b)) if num_ok(b) else T._node("div", a, b)
            if T.const[a] and num_ok(a) and not shared:
                return T._node("div", a, b)
            return T.mul(a, T.recip(b))
        cval_memo, bnd_memo = {}, {}

        def cvals(node, T=T):
            """per-unit values of a constant template node"""
            if node not in cval_memo:
                v = np.asarray(_np_eval(T, [node], const_leaf)[node], dtype=np.float64)
                cval_memo[node] = np.broadcast_to(v, (n,))
            return cval_memo[node]

        def cval(node, T=T):
            """the value of a constant template node that is the same for every unit, else None"""
            if not T.const[node]:
                return None
            v = cvals(node)
            return float(v[0]) if np.all(v == v[0]) else None

        def bounds(node, T=T):
            """(lo, hi, never NaN) of a template node over all units and all positions: what lets
            0 * x be folded to 0. Conservative: anything not proven is (-inf, inf, False). fmax / fmin
            return the other operand for a NaN (IEEE maxNum, the emitted fmax / fmin and v_max_f64)."""
            if node in bnd_memo:
                return bnd_memo[node]
            inf, unknown = math.inf, (-math.inf, math.inf, False)
            op = T.ops[node]
            k, r = op[0], unknown
            if T.const[node]:
                v = cvals(node)
                r = (float(np.min(v)), float(np.max(v)), True) if np.all(np.isfinite(v)) else unknown
            elif k in ("gat", "ext", "q", "red", "wred"):
                r = unknown
            else:
                b = [bounds(a) for a in op[1:]]
                fin = lambda x: x[2] and math.isfinite(x[0]) and math.isfinite(x[1])   # noqa: E731
                with np.errstate(all="ignore"):
                    if k == "neg":
                        r = (-b[0][1], -b[0][0], b[0][2])
                    elif k in ("add", "sub") and fin(b[0]) and fin(b[1]):
                        r = ((b[0][0] + b[1][0], b[0][1] + b[1][1], True) if k == "add"
                             else (b[0][0] - b[1][1], b[0][1] - b[1][0], True))
                    elif k == "mul" and fin(b[0]) and fin(b[1]):
                        c = [x * y for x in b[0][:2] for y in b[1][:2]]
                        r = (min(c), max(c), True)
                    elif k == "div" and fin(b[0]) and fin(b[1]) and (b[1][0] > 0.0 or b[1][1] < 0.0):
                        c = [x / y for x in b[0][:2] for y in b[1][:2]]
                        r = (min(c), max(c), True)
                    elif k == "exp" and b[0][2] and b[0][1] < 700.0:
                        r = (float(np.exp(b[0][0])), float(np.exp(b[0][1])), True)
                    elif k == "log" and fin(b[0]) and b[0][0] > 0.0:
                        r = (float(np.log(b[0][0])), float(np.log(b[0][1])), True)
                    elif k == "log1p" and fin(b[0]) and b[0][0] > -1.0:
                        r = (float(np.log1p(b[0][0])), float(np.log1p(b[0][1])), True)
                    elif k == "abs" and b[0][2]:
                        r = (0.0, max(abs(b[0][0]), abs(b[0][1])), True)
                    elif k == "erf":
                        r = (-1.0, 1.0, b[0][2])
                    elif k in ("max", "min") and (b[0][2] or b[1][2]):
                        los = [x[0] for x in b if x[2]]
                        his = [x[1] if x[2] else inf for x in b]
                        lo_all = [x[0] if x[2] else -inf for x in b]
                        r = ((max(los), max(his), True) if k == "max" else (min(lo_all), min(x[1] for x in b if x[2]), True))
                    elif k == "sel_gt":
                        r = (min(b[2][0], b[3][0]), max(b[2][1], b[3][1]), b[2][2] and b[3][2])
                if not (r[0] == r[0] and r[1] == r[1]):          # a NaN bound proves nothing
                    r = unknown
            bnd_memo[node] = r
            return r

        def finite(node):
            lo, hi, ok = bounds(node)
            return ok and math.isfinite(lo) and math.isfinite(hi)

        def simple(node, T=T):
            """a constant that is 0, 1 or -1 in every unit as a literal (so that the exact rewrites of
            the graph's own mul / neg apply to it in the adjoint pass too)"""
            c = cval(node)
            return T.lit(c) if c in (0.0, 1.0, -1.0) and T.ops[node][0] != "lit" else node

        def make(kind, a, T=T):
            """a template node with the rewrites a constant factor or summand allows: 1 * x, x + 0 and
            0 * x for an x that is finite whatever the position (bounds)."""
            if kind == "mul":
                for x, y in ((a[0], a[1]), (a[1], a[0])):
                    c = cval(x)
                    if c == 0.0 and finite(y):
                        return T.lit(0.0)
                    if c == 1.0:
                        return y
                    if c == -1.0:
                        return T.neg(y)
            elif kind == "add":
                if cval(a[0]) == 0.0:
                    return a[1]
                if cval(a[1]) == 0.0:
                    return a[0]
            elif kind == "sub":
                if cval(a[1]) == 0.0:
                    return a[0]
                if cval(a[0]) == 0.0:
                    return T.neg(a[1])
            return simple(T._node(kind, *a))
        for k in sorted(need):
            node0 = f.ids[0][k]
            kind = shape[k][0]
            if kind == "lit":
                tmap[k] = T._node("lit", shape[k][1])
            elif kind == "wlit" and f.uniform[k]:
                tmap[k] = T._node("lit", g.ops[node0][1])
            elif f.uniform[k]:
                if g.const[node0]:
                    tmap[k] = T._node("uc", uc_slot(("node", node0), g_const_value(node0)))
                else:
                    if node0 not in b_index:
                        b_index[node0] = len(boundary)
                        boundary.append(node0)
                    tmap[k] = T._node("ext", b_index[node0])
            elif kind == "q":
                f.gather.append([g.ops[ids[k]][1] for ids in f.ids])
                tmap[k] = T._node("gat", len(f.gather) - 1)
            elif kind in ("data", "wlit"):
                vals = [g.data[g.ops[ids[k]][1]] if kind == "data" else float.fromhex(g.ops[ids[k]][1])
                        for ids in f.ids]
                if all(v == vals[0] for v in vals):
                    tmap[k] = (T.lit(vals[0]) if vals[0] in (0.0, 1.0, -1.0)
                               else T._node("uc", uc_slot(("val", float(vals[0]).hex()), vals[0])))
                else:
                    f.raw_cols.append(np.asarray(vals, dtype=np.float64))
                    tmap[k] = T._node("col", -len(f.raw_cols))      # raw columns: negative ids
            elif kind == "div":
                tmap[k] = quotient(tmap[shape[k][1]], tmap[shape[k][2]], shape[k][2] in share_recip)
            else:
                tmap[k] = make(kind, [tmap[a] for a in shape[k][1:]])
        f.T, f.troot = T, tmap[root]
        n_fwd = len(T.ops)
        ad = cg._Grad(T, f.troot, const_num_ok=num_ok)
        ad.run(n_fwd)
        f.ext_adj = {}          # boundary index -> adjoint node
        for key, node in list(T.key.items()):
            if key[0] == "ext" and ad.adj.get(node) is not None:
                f.ext_adj[key[1]] = ad.adj[node]
        f.gat_adj = [ad.adj.get(T.key[("gat", p)]) for p in range(len(f.gather))]
        # fold what depends on constants only (per-unit data, uniform constants) into columns
        outputs = [f.troot] + list(f.ext_adj.values()) + [a for a in f.gat_adj if a is not None]
        live, stack = set(), list(outputs)
        while stack:
            i = stack.pop()
            if i in live:
                continue
            live.add(i)
            if T.ops[i][0] not in _LEAVES and not T.const[i]:
                stack.extend(T.ops[i][1:])
        fold = sorted(i for i in live if T.const[i] and T.ops[i][0] not in ("lit", "uc"))
        raw = f.raw_cols

        def leaf(op, raw=raw):
            if op[0] == "col":
                return raw[-op[1] - 1]
            if op[0] == "uc":
                return np.float64(uc_vals[op[1]])
            raise cg.CodegenError("unexpected leaf %r in a constant" % (op,))
        vals = _np_eval(T, fold, leaf)
        f.cols, f.col_of = [], {}
        for i in fold:
            v = np.broadcast_to(np.asarray(vals[i], dtype=np.float64), (n,))
            if np.all(v == v[0]):
                f.col_of[i] = ("uc", uc_slot(("val", float(v[0]).hex()), v[0]))
            else:
                f.col_of[i] = ("col", len(f.cols))
                f.cols.append(np.array(v))
        f.live, f.tmap = live, tmap

    for f in families:
        build_template(f)
    lsh = sh_off

    # ---- the uniform part: scalar units + the boundary nodes, differentiated with the reduced
    # adjoints as seeds (U = scalar_lp + sum_j red_j * b_j) ----
    scalar_lp = None
    for t in scalar_units:
        scalar_lp = t if scalar_lp is None else g.add(t, scalar_lp)
    if scalar_lp is None:
        scalar_lp = g.lit(0.0)

    # ---- families INSIDE the uniform part: a sum of like terms of shared values only (the eight
    # quotients of a Lanczos series, math.ex:27-52) is evaluated one term per lane and reduced in a
    # butterfly of its own before the family loops, instead of by every lane in full. Per term the
    # value and its partial derivatives with respect to the term's uniform inputs; the tangent of the
    # sum is sum_e (reduced partial e) * tangent(e). One level: a sum whose inputs need another
    # spread sum stays where it is. ----
    ufams = _uniform_families(g, [scalar_lp] + list(boundary))
    for f in ufams:
        build_template(f)
        if f.gather:
            raise cg.CodegenError("internal: a uniform family gathered a variable")
    uf_of = {f.out: f for f in ufams}
    NW, w_const = 0, {}
    for f in ufams:
        f.w0 = NW
        f.w_of = {j: f.w0 + 1 + k for k, j in enumerate(sorted(f.ext_adj))}
        NW += 1 + len(f.ext_adj)
        cpart = np.float64(0.0)
        for c in f.const_part:                 # folded left to right, added after the reduction
            cpart = cpart + np.float64(g_const_value(c))
        f.cpart = float(cpart)
    n_split = len(g.ops)
    # reduced values: s[0] = log-density of the families, s[1 + j] = adjoint of boundary node j
    acc_of = {}
    for f in families:
        for j in sorted(f.ext_adj):
            if j not in acc_of:
                acc_of[j] = 1 + len(acc_of)
    NS = 1 + len(acc_of)
    # Forward mode for the uniform part: it has few inputs (the shared variables) and is evaluated
    # BEFORE the family loops, while its reduced adjoint seeds only exist after them -- a reverse
    # sweep would keep every intermediate of e.g. two Lanczos series alive across the loops (a
    # hundred vector registers in the sampling kernel). Tangents d node / d shared variable are
    # computed next to the values instead; what stays live is the Jacobian of the boundary nodes:
    #   d logp / d q_i = d scalar_lp / d q_i + sum_j red_j * d b_j / d q_i
    fwd_roots = [scalar_lp] + list(boundary)
    spread = {f.out: [(boundary[j], g._node("wred", f.w_of[j])) for j in sorted(f.ext_adj)] for f in ufams}
    tan = _forward_tangents(g, fwd_roots, spread)
    ug = {}
    for i in range(D):
        acc = tan.get(scalar_lp, {}).get(i)
        for j, sl_ in sorted(acc_of.items(), key=lambda kv: kv[1]):
            tj = tan.get(boundary[j], {}).get(i)
            if tj is not None:
                term = g.mul(g._node("red", sl_), tj)
                acc = term if acc is None else g.add(acc, term)
        if acc is not None:
            ug[i] = acc

    # liveness of the uniform graph
    outputs = [scalar_lp] + list(ug.values()) + [b for b in boundary]
    live, stack = set(), list(outputs)
    while stack:
        i = stack.pop()
        if i in live:
            continue
        live.add(i)
        if i in uf_of:
            stack.extend(uf_of[i].inputs)       # its terms are evaluated by the lanes, not here
        elif g.ops[i][0] not in _LEAVES and not g.const[i]:
            stack.extend(g.ops[i][1:])
    for i in sorted(live):
        if g.const[i] and g.ops[i][0] != "lit":
            uc_slot(("node", i), g_const_value(i))
    for f in ufams:
        if f.const_part:
            uc_slot(("val", f.cpart.hex()), f.cpart)
    # nodes that need a spread sum come after its butterfly
    after_w = set()
    for i in sorted(live):
        if i in uf_of or (not g.const[i] and g.ops[i][0] not in _LEAVES and any(a in after_w for a in g.ops[i][1:])):
            after_w.add(i)

    # ---- gather lists of the owner lanes (padded to the widest lane per slot) ----
    contrib = [[] for _ in range(D)]
    for f in families:
        for p in range(len(f.gather)):
            if f.strip[p] < 0:
                continue
            for u, var in enumerate(f.gather[p]):
                contrib[var].append(f.strip[p] + u)
    width = [0] * DPL
    for i in range(D):
        width[i // G] = max(width[i // G], len(contrib[i]))
    zero_cell = D
    # per lane contiguous: [lane][slot k][j < width[k]] -- a lane's whole list is NELL ints, which the
    # device functor keeps in registers when it is short (Lane::ell)
    ell_off, NELL = [], 0
    for k in range(DPL):
        ell_off.append(NELL)
        NELL += width[k]
    ell = []
    for l in range(G):
        for k in range(DPL):
            i = l + k * G
            for j in range(width[k]):
                ell.append(contrib[i][j] if (i < D and j < len(contrib[i])) else zero_cell)

    # ---- table layout: [uc][double columns][int32 columns (gather indices, owner lists)] ----
    # Per-unit columns in PAIRS since round 6: columns 2p and 2p + 1 of a family are interleaved over its units
    # (unit u's two entries at doff + 2 p npad + 2 u, + 1; an odd last column alone, one double per unit), the first
    # pair on a 128-byte boundary of the table. A unit's pair is one 16-byte load, and the G lanes of a chain --
    # and the 64 / G chains of a wavefront, which read the same units -- read 16 G CONSECUTIVE bytes per load:
    # whole cache lines from L2, and no bank conflict when the table sits in an LDS image. Row-major (unit u's
    # columns consecutive: a stride of 8 ncol bytes between lanes) cost the generated 500 x 20 regression 21
    # bank-conflict cycles per LDS read in the workgroup form (profiles/r6_gen_wg).
    NUC = max(1, len(uc_vals))
    NUC += (-NUC) % 16
    doff = NUC
    for f in families + ufams:
        f.doff = doff
        nc_ = len(f.cols)
        # (offset, stride in doubles) of column c. Short rows stay row-major (unit u's columns consecutive): one or
        # two loads per unit either way, and the lone-wave kernels of radon / sv measured 3-7 % slower in pairs
        f.paired = nc_ >= PAIR_MIN_COLS
        if f.paired:
            f.cpos = [(doff + (c // 2) * 2 * f.npad + (c & 1), 2) if c < nc_ - (nc_ & 1)
                      else (doff + (nc_ - 1) * f.npad, 1) for c in range(nc_)]
        else:
            f.cpos = [(doff + c, nc_) for c in range(nc_)]
        doff += nc_ * f.npad
    ints = []
    for f in families:
        f.ioff = len(ints)
        for p in range(len(f.gather)):
            col = list(f.gather[p]) + [f.gather[p][0]] * (f.npad - f.n)
            ints.extend(col)
    ell_base = len(ints)
    ints.extend(ell)
    if len(ints) % 2:
        ints.append(0)
    dtab = [np.asarray(uc_vals + [0.0] * (NUC - len(uc_vals)), dtype=np.float64)]
    for f in families + ufams:
        if f.cols:
            rows = np.stack(f.cols, axis=1)                       # [n][ncol]
            rows = np.concatenate([rows, np.repeat(rows[:1], f.npad - f.n, axis=0)])
            nc_ = rows.shape[1]
            if not f.paired:
                dtab.append(rows.ravel())                                               # [npad][ncol]
                continue
            for p_ in range(nc_ // 2):
                dtab.append(np.ascontiguousarray(rows[:, 2 * p_:2 * p_ + 2]).ravel())   # [npad][2]
            if nc_ & 1:
                dtab.append(np.ascontiguousarray(rows[:, nc_ - 1]))                     # [npad]
    data = np.concatenate(dtab + [np.asarray(ints, dtype=np.int32).view(np.float64)]) \
        if ints else np.concatenate(dtab)
    ioff_doubles = doff

    # ---- emission ----
    def lit_text(hexv):
        s = repr(float.fromhex(hexv))
        if "inf" in s or "nan" in s:
            raise cg.CodegenError("non-finite literal")
        return "(%s)" % s if s.startswith("-") else s

    fn1 = {k: v.replace("EXMC_GEN_", "EXMC_GENL_") for k, v in cg._FN1.items()}
    fn1["abs"] = "fabs"

    def expr(op, a):
        if op in cg._BIN:
            return "%s %s %s" % (a[0], cg._BIN[op], a[1])
        if op == "neg":
            return "-%s" % a[0]
        if op in fn1:
            return "%s(%s)" % (fn1[op], a[0])
        if op in cg._FN2:
            return "%s(%s, %s)" % (cg._FN2[op], a[0], a[1])
        if op == "sel_gt":
            return "(%s > %s) ? %s : %s" % tuple(a)
        raise cg.CodegenError("cannot emit %s" % op)

    def fuse_plan(gr, nodes, pinned):
        """Contraction at emission (the lane layout's own numeric contract, like the fma chains of the
        hand-written kernels): a product with a single consumer that is a sum or a difference is
        emitted as one fused multiply-add. -> {consumer: (form, product)}, set of absorbed products."""
        uses = {}
        for i in nodes:
            for a in gr.ops[i][1:]:
                uses[a] = uses.get(a, 0) + 1
        plan, gone = {}, set()
        for i in nodes:
            op = gr.ops[i]
            if op[0] not in ("add", "sub"):
                continue
            for side in (1, 0):
                m = op[1 + side]
                if (m in nodes and gr.ops[m][0] == "mul" and uses.get(m, 0) == 1 and m not in pinned
                        and m not in gone and not gr.const[m]):
                    plan[i] = (op[0], side, m)
                    gone.add(m)
                    break
        # an output that is a product nobody else reads is fused into its accumulation
        acc_mul = set(m for m in pinned if m in nodes and gr.ops[m][0] == "mul" and uses.get(m, 0) == 0
                      and not gr.const[m])
        return plan, gone, acc_mul

    def fused(gr, i, plan, ref):
        kind, side, m = plan[i]
        a, b = (ref(x) for x in gr.ops[m][1:])
        other = ref(gr.ops[i][2 - side])
        if kind == "add":
            return "EXMC_GEN_FMA(%s, %s, %s)" % (a, b, other)
        if side == 1:                               # other - a * b
            return "EXMC_GEN_FMA(-(%s), %s, %s)" % (a, b, other)
        return "EXMC_GEN_FMA(%s, %s, -(%s))" % (a, b, other)     # a * b - other

    def uref(i):
        op = g.ops[i]
        if op[0] == "lit":
            return lit_text(op[1])
        if op[0] == "red":
            return "s[%d]" % op[1]
        if op[0] == "wred":
            return "w[%d]" % op[1]
        if g.const[i]:
            return "EXMC_GEN_LT(%d)" % uc_of[("node", i)]
        return "u%d" % i

    u_nodes = set(i for i in live if not g.const[i] and g.ops[i][0] not in _LEAVES and i not in uf_of)
    u_pinned = set([scalar_lp] + list(ug.values()) + list(boundary))
    u_plan, u_gone, _ = fuse_plan(g, u_nodes, u_pinned)

    # Chain-scalar transcendentals of the uniform part are the same instruction stream whatever their
    # argument: the logs (exps, reciprocals) of one dependency level are evaluated together, argument
    # i by lane i of the group, and broadcast back (exmc_device.hpp lane_batch) -- each value by
    # exactly the operations of the plain call, so the host checker's loop gives the same bits.
    batch_names = {"log": "LOG", "exp": "EXP", "log1p": "LOG1P", "rcp": "RCP"}
    n_batches = [0]

    def batch_kind(i):
        op = g.ops[i]
        if op[0] in ("log", "exp", "log1p"):
            return op[0]
        if op[0] == "div" and g.lit_value(op[1]) == 1.0:
            return "rcp"
        return None

    def ustmts(lo, hi, late=None):
        out = []
        if lo == 0 and not late:     # the shared variables: broadcast reads of the position strip
            out.extend("  const double u%d = EXMC_GEN_SH(%d);" % (i, g.ops[i][1])
                       for i in sorted(live) if g.ops[i][0] == "q")
        region = [i for i in sorted(u_nodes)
                  if lo <= i < hi and (late is None or (i in after_w) == late)]
        level = {}
        for i in region:
            lv = max([level.get(a, 0) for a in g.ops[i][1:]] or [0])
            level[i] = lv + (1 if batch_kind(i) else 0)
        for lv in range(0, max(level.values(), default=0) + 1):
            groups = {}
            for i in region:
                if level[i] == lv and batch_kind(i):
                    groups.setdefault(batch_kind(i), []).append(i)
            for kind in sorted(groups):
                ids = groups[kind]
                for c0 in range(0, len(ids), min(G, 16)):
                    chunk = ids[c0:c0 + min(G, 16)]
                    args = [uref(g.ops[i][2] if kind == "rcp" else g.ops[i][1]) for i in chunk]
                    if len(chunk) == 1:
                        i = chunk[0]
                        out.append("  const double u%d = %s;" % (i, expr(g.ops[i][0], [uref(x) for x in g.ops[i][1:]])))
                        continue
                    b = "b%d" % n_batches[0]
                    n_batches[0] += 1
                    out.append("  double %s[%d] = {%s};" % (b, len(chunk), ", ".join(args)))
                    out.append("  EXMC_GEN_BATCH_%s(%d, %s);" % (batch_names[kind], len(chunk), b))
                    out.extend("  const double u%d = %s[%d];" % (i, b, j) for j, i in enumerate(chunk))
            for i in region:
                if level[i] == lv and not batch_kind(i) and i not in u_gone:
                    op = g.ops[i]
                    e = fused(g, i, u_plan, uref) if i in u_plan else expr(op[0], [uref(x) for x in op[1:]])
                    out.append("  const double u%d = %s;" % (i, e))
        return out

    L = []
    L.append("/* lane layout (exmc_amd/codegen_lanes.py): %d lanes per chain, %d dimensions per lane;"
             % (G, DPL))
    L.append(" * %d famil%s of repeated terms (%s units), %d uniform term%s. lt = [%d uniform constants]"
             % (len(families), "y" if len(families) == 1 else "ies",
                " + ".join(str(f.n) for f in families) or "0", len(scalar_units),
                "" if len(scalar_units) == 1 else "s", NUC))
    L.append(" * [per-unit columns][int32: gather indices, owner lists]; EXMC_GEN_SH(i) = double i of the")
    L.append(" * chain's LDS strip: [position (d)][0.0][adjoint strips]. */")
    L.append("#define EXMC_GEN_LANES %d" % G)
    L.append("#define EXMC_GEN_DPL %d" % DPL)
    L.append("#define EXMC_GEN_LSH %d" % lsh)
    L.append("#define EXMC_GEN_NS %d" % NS)
    L.append("#define EXMC_GEN_NW %d   /* sums of the spread part of the uniform terms (0: none) */" % NW)
    L.append("#define EXMC_GEN_NLT %d" % data.size)
    L.append("#define EXMC_GEN_NELL %d   /* ints of a lane's owner list */" % max(1, NELL))
    L.append("#define EXMC_GEN_ELL_OFF %d   /* ... of lane l at ((const int*)lt)[EXMC_GEN_ELL_OFF + l * EXMC_GEN_NELL] */"
             % (2 * ioff_doubles + ell_base))
    L.append("#define EXMC_GEN_WAVES_PER_SIMD %d" % waves_per_simd)
    # the sampling kernel as workgroups of eight wavefronts around ONE LDS image of the tables (exmc_nuts.hpp
    # nuts_kernel_wg): for a layout of several chains per wavefront with two waves per SIMD whose tables are too
    # large to sit beside a one-wave workgroup (exmc_models.hpp EXMC_GEN_TABLE_IN_LDS) and fit beside eight tree
    # stacks, the ziggurat tables and eight sets of strips in a compute unit's 160 KB
    wg_lds = 8 * ((5 * DPL + 3) * 64 * 8 + (64 // G) * lsh * 8) + 768 * 8 + 8 + int(data.size) * 8
    wg = int(G < 64 and waves_per_simd == 2 and data.size > 2048 and wg_lds <= 160 * 1024)
    L.append("#define EXMC_GEN_WG %d   /* 1: the plug-in carries the workgroup form of the sampling kernel too */" % wg)
    L.append("")
    L.append("#define EXMC_GEN_IOFF %d   /* the int32 tables start at double EXMC_GEN_IOFF of lt */" % ioff_doubles)
    L.append("")
    L.append("#else   /* EXMC_GEN_LANES_SECTION: the lane function itself. Included once per table placement")
    L.append("       * with EXMC_GEN_LANES_NAME, EXMC_GEN_LT(i) (double i of the table) and EXMC_GEN_IT(i) (int32 i")
    L.append("       * of its index part) defined by the includer: global memory, or an LDS image of it */")
    L.append("EXMC_GEN_FN double EXMC_GEN_LANES_NAME(const double* lt, const int* el, int l, double* g EXMC_GEN_CTX_DECL) {")
    L.append("  EXMC_GEN_SH(%d) = 0.0;" % zero_cell)
    def emit_family(f, title, arr, slot_of, tag, split=False):
        T = f.T
        t_nodes = set(i for i in f.live if not T.const[i] and T.ops[i][0] not in _LEAVES)
        # what does not change from unit to unit (a function of uniform values and uniform constants
        # only: the reciprocal of a shared scale) is evaluated once, in front of the loop
        fixed = {}
        for i in sorted(f.live):
            op = T.ops[i]
            if op[0] in ("lit", "uc", "ext"):
                fixed[i] = True
            elif op[0] in ("col", "gat"):
                fixed[i] = False
            elif i in f.col_of:
                fixed[i] = f.col_of[i][0] == "uc"
            else:
                fixed[i] = all(fixed.get(a, False) for a in op[1:])
        hoisted = set(i for i in t_nodes if fixed[i])

        def tref(i, T=T, f=f):
            op = T.ops[i]
            if op[0] == "lit":
                return lit_text(op[1])
            if op[0] == "uc":
                return "EXMC_GEN_LT(%d)" % op[1]
            if i in f.col_of:
                kind, k = f.col_of[i]
                return "EXMC_GEN_LT(%d)" % k if kind == "uc" else "c%d" % k
            if op[0] == "ext":
                return uref(boundary[op[1]])
            if op[0] == "gat":
                return "v%d" % op[1]
            return ("%s_%d" % (tag, i)) if i in hoisted else ("t%d" % i)
        t_pinned = set([f.troot] + list(f.ext_adj.values()) + [a for a in f.gat_adj if a is not None])
        t_plan, t_gone, t_accmul = fuse_plan(T, t_nodes, t_pinned)
        L.append("  /* %s: %d units, %d per lane */" % (title, f.n, f.S))
        for i in sorted(hoisted):
            if i in t_gone:
                continue
            op = T.ops[i]
            e = fused(T, i, t_plan, tref) if i in t_plan else expr(op[0], [tref(x) for x in op[1:]])
            L.append("  const double %s_%d = %s;" % (tag, i, e))
        g0, ng = ("EXMC_GEN_G0", "EXMC_GEN_NG") if split else ("0", "1")   # (the groups of the wavefront
        nc, ngat = len(f.cols), len(f.gather)                                #  share a family in the one-chain warmup)
        # A unit's table row and gathered variables are loads the unit's arithmetic waits for, and a
        # lone wave per SIMD has nothing else to issue meanwhile (radon: 15 slots x ~800 clocks of L2
        # latency per leapfrog against ~900 vector instructions). Short rows are therefore fetched
        # for a block of slots at once -- the indices, then the variables, then the arithmetic.
        B = max(1, min(f.S, PREFETCH_DOUBLES[waves_per_simd] // max(1, nc + ngat)))
        blocked = B > 1 and (nc + ngat) > 0
        if blocked:
            L.append("  for (int jb = 0; %s + jb * %s < %d; jb += %d) {" % (g0, ng, f.S, B))
            L.append("    int un_[%d];" % B)
            if nc:
                L.append("    double c_[%d][%d];" % (B, nc))
            if ngat:
                L.append("    int ix_[%d][%d];" % (B, ngat))
                L.append("    double v_[%d][%d];" % (B, ngat))
            L.append("    for (int j = 0; j < %d; j++) {" % B)
            L.append("      const int sl = %s + (jb + j) * %s;" % (g0, ng))
            L.append("      const int un = sl * %d + l;" % G)
            L.append("      un_[j] = (sl < %d && un < %d) ? un : -1;" % (f.S, f.n))
            L.append("      const int uc = un_[j] < 0 ? 0 : un;")
            for c in range(nc):
                if f.paired and c + 1 < nc and not (c & 1):   # a pair: ONE 16-byte load (EXMC_GEN_LT2: aligned)
                    L.append("      { const exmc_gen_d2 cc_ = EXMC_GEN_LT2(%d + uc * 2); c_[j][%d] = cc_.x; c_[j][%d] = cc_.y; }"
                             % (f.cpos[c][0], c, c + 1))
                elif not (f.paired and (c & 1)):
                    L.append("      c_[j][%d] = EXMC_GEN_LT(%d + uc * %d);" % (c, f.cpos[c][0], f.cpos[c][1]))
            for p in range(ngat):
                L.append("      ix_[j][%d] = EXMC_GEN_IT(%d + uc);" % (p, f.ioff + p * f.npad))
            L.append("    }")
            if ngat:
                L.append("    for (int j = 0; j < %d; j++) {" % B)
                for p in range(ngat):
                    L.append("      v_[j][%d] = EXMC_GEN_SH(ix_[j][%d]);" % (p, p))
                L.append("    }")
            L.append("    for (int j = 0; j < %d; j++) {" % B)
            L.append("    const int un = un_[j];")
            L.append("    if (un >= 0) {")
            for p in range(ngat):
                L.append("    const double v%d = v_[j][%d];" % (p, p))
            for c in range(nc):
                L.append("    const double c%d = c_[j][%d];" % (c, c))
        else:
            L.append("  for (int sl = %s; sl < %d; sl += %s) {" % (g0, f.S, ng))
            L.append("    const int un = sl * %d + l;" % G)
            if f.n < f.npad:
                L.append("    if (un < %d) {" % f.n)
            for p in range(ngat):
                L.append("    const double v%d = EXMC_GEN_SH(EXMC_GEN_IT(%d + un));" % (p, f.ioff + p * f.npad))
            for c in range(nc):
                if f.paired and c + 1 < nc and not (c & 1):   # a pair: ONE 16-byte load (EXMC_GEN_LT2: aligned)
                    L.append("    const exmc_gen_d2 cc%d = EXMC_GEN_LT2(%d + un * 2);" % (c, f.cpos[c][0]))
                    L.append("    const double c%d = cc%d.x;" % (c, c))
                    L.append("    const double c%d = cc%d.y;" % (c + 1, c))
                elif not (f.paired and (c & 1)):
                    L.append("    const double c%d = EXMC_GEN_LT(%d + un * %d);" % (c, f.cpos[c][0], f.cpos[c][1]))
        n_use = {}
        for x in [f.troot] + list(f.ext_adj.values()):
            n_use[x] = n_use.get(x, 0) + 1
        t_accmul = set(m for m in t_accmul if n_use.get(m, 0) == 1 and m not in [a for a in f.gat_adj if a is not None])

        def accumulate(slot, node, T=T, t_accmul=t_accmul, tref=tref):
            if node in t_accmul:
                a, b = (tref(x) for x in T.ops[node][1:])
                return "    %s[%d] = EXMC_GEN_FMA(%s, %s, %s[%d]);" % (arr, slot, a, b, arr, slot)
            return "    %s[%d] = %s[%d] + %s;" % (arr, slot, arr, slot, tref(node))
        for i in sorted(t_nodes):
            if i in t_gone or i in t_accmul or i in hoisted:
                continue
            op = T.ops[i]
            e = fused(T, i, t_plan, tref) if i in t_plan else expr(op[0], [tref(x) for x in op[1:]])
            L.append("    const double t%d = %s;" % (i, e))
        L.append(accumulate(slot_of[None], f.troot))
        for j in sorted(f.ext_adj):
            L.append(accumulate(slot_of[j], f.ext_adj[j]))
        for p in range(len(f.gather)):
            if f.strip[p] >= 0:
                L.append("    EXMC_GEN_SH(%d + un) = %s;" % (f.strip[p], tref(f.gat_adj[p])))
        if blocked:
            L.append("    }")
            L.append("    }")
        elif f.n < f.npad:
            L.append("    }")
        L.append("  }")

    L.extend(ustmts(0, n_split, late=False))
    if ufams:
        L.append("  double w[EXMC_GEN_NW];")
        L.append("  for (int j = 0; j < EXMC_GEN_NW; j++) w[j] = 0.0;")
        for fi, f in enumerate(ufams):
            slots = dict(f.w_of)
            slots[None] = f.w0
            emit_family(f, "spread sum %d of the uniform part" % fi, "w", slots, "hw%d" % fi)
        L.append("  EXMC_GEN_ALLSUM_W(w);")
        for f in ufams:
            if f.const_part:
                L.append("  const double u%d = w[%d] + EXMC_GEN_LT(%d);" % (f.out, f.w0, uc_of[("val", f.cpart.hex())]))
            else:
                L.append("  const double u%d = w[%d];" % (f.out, f.w0))
        L.extend(ustmts(0, n_split, late=True))
    L.append("  double s[EXMC_GEN_NS];")
    L.append("  for (int j = 0; j < EXMC_GEN_NS; j++) s[j] = 0.0;")
    for fi, f in enumerate(families):
        slots = dict(acc_of)
        slots[None] = 0
        emit_family(f, "family %d" % fi, "s", slots, "hf%d" % fi, split=True)
    L.append("  EXMC_GEN_ALLSUM(s);")
    L.append("  EXMC_GEN_XGROUP(s);   /* EXMC_GEN_NG > 1: the groups' sums, group 0 first */")
    L.extend(ustmts(n_split, len(g.ops)))
    L.append("  EXMC_GEN_FENCE();")
    for k in range(DPL):
        L.append("  {")
        L.append("    const int dim = l + %d;" % (k * G))
        L.append("    double acc = 0.0;")
        if width[k] > 0:
            L.append("    for (int j = 0; j < %d; j++) acc = acc + EXMC_GEN_SH(el[%d + j]);" % (width[k], ell_off[k]))
        sel = "0.0"
        for i in sorted(ug, reverse=True):
            if i // G == k:
                sel = "(dim == %d) ? %s : (%s)" % (i, uref(ug[i]), sel)
        L.append("    const double ugs = %s;" % sel)
        L.append("    g[%d] = acc + ugs;" % k)
        L.append("    (void)dim;")
        L.append("  }")
    L.append("  (void)el; (void)lt;")
    L.append("  return %s + s[0];" % uref(scalar_lp))
    L.append("}")
    L.append("#endif")
    text = "\n".join(L) + "\n"
    return dict(text=text, data=data, lanes=G, dpl=DPL, lsh=lsh, n_families=len(families),
                family_sizes=[f.n for f in families], n_scalar_units=len(scalar_units),
                spread_sizes=[f.n for f in ufams], n_spread_sums=NW, n_batches=n_batches[0],
                n_reduced=NS, n_boundary=len(boundary), gather_width=width, wg=wg)
