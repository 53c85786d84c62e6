"""Plates across lanes: the 16-lane layout for generated models (exmc_amd/codegen.py).

One lane per chain keeps 15/16 of the chip idle at a few thousand chains. The hand-written
eight_schools kernel spreads a chain over 16 lanes: lane j owns school j's parameter, the eight
likelihood terms are ONE instruction stream over eight lanes, and three per-chain sums (log-density,
d/dmu, d/dtau) go through one DPP butterfly. This module finds that structure in a Builder IR:

  * every logp term (a free RV's prior term, an obs node, or one element of a vector obs) is a
    *unit*; units whose expression trees are identical up to their leaves form a *family*;
  * a family is vectorised when, position by position, its free-variable leaves are either the
    same variable in every unit (shared, e.g. mu, tau) or a different variable in every unit
    (private, e.g. theta_j; at most one private variable per unit). The unit whose private
    variable is dimension i runs on lane i, which owns that dimension (D <= 16, one dimension per
    lane); units without a private variable take lanes 0, 1, ...; per-unit constants (y_j,
    sigma_j, ...) become a per-lane data row;
  * everything else (single terms such as the hyper-priors) is evaluated by every lane redundantly.

Per lane the generated `exmc_gen_lane` returns: the lane's partial log-density of the vectorised
families and its partial adjoints of the shared variables (-> one `group_allsum_n`), the adjoint of
its own private variable, and the scalar part's value and adjoints (identical on every lane). The
numeric contract is the lane contract of the hand-written kernels: per-lane
partials, then the xor butterfly; the host checker runs the same text in a loop over 16 virtual
lanes and adds in that order.
"""
import numpy as np

from . import codegen as cg

G = 16


class _VGraph(cg._Graph):
    """_Graph with two more leaves: `qown` (the lane's own dimension) and `ldata k` (column k of
    the lane's data row). const[] stays "not dynamic" (what _Grad skips); lane[] marks constants
    that differ from lane to lane."""

    LEAF_CONST = {"lit": True, "data": True, "q": False, "qown": False, "ldata": True,
                  "qleaf": False, "dleaf": True}

    def __init__(self):
        super().__init__()
        self.lane = []

    def _node(self, op, *args):
        k = (op,) + args
        i = self.key.get(k)
        if i is None:
            i = len(self.ops)
            self.ops.append(k)
            if op in self.LEAF_CONST:
                c, ln = self.LEAF_CONST[op], op == "ldata"
            else:
                c = all(self.const[a] for a in args)
                ln = c and any(self.lane[a] for a in args)
            self.const.append(c)
            self.lane.append(ln)
            self.key[k] = i
        return i

    def qown(self):
        return self._node("qown")

    def ldata(self, k):
        return self._node("ldata", k)


class _Rec:
    """Leaf provider that records a unit's leaves in order of use (for the family signature)."""

    def __init__(self):
        self.g = _VGraph()
        self.qs, self.ds = [], []

    def q(self, i):
        self.qs.append(i)
        return self.g._node("qleaf", len(self.qs) - 1)

    def datum(self, x):
        self.ds.append(float(x))
        return self.g._node("dleaf", len(self.ds) - 1)


class _Plain:
    def __init__(self, g):
        self.g = g

    def q(self, i):
        return self.g.q(i)

    def datum(self, x):
        return self.g.datum(x)


class _Template:
    """Leaf provider for a family's template: shared variables and constants that are equal in
    every unit become ordinary leaves, the rest the lane's own dimension / data columns."""

    def __init__(self, g, qplan, dplan):
        self.g, self.qplan, self.dplan = g, qplan, dplan
        self.qi = self.di = 0

    def q(self, _i):
        kind, idx = self.qplan[self.qi]
        self.qi += 1
        return self.g.q(idx) if kind == "shared" else self.g.qown()

    def datum(self, _x):
        kind, v = self.dplan[self.di]
        self.di += 1
        return self.g.datum(v) if kind == "uniform" else self.g.ldata(v)


def _units(nodes, ncp_info, offset):
    """The term walk of codegen.generate, one closure per unit: build(leaves) -> node."""
    def resolve_ref(L, id_, stack=()):
        if id_ not in offset:
            raise cg.CodegenError("param ref %r is not a free random variable" % id_)
        if id_ in stack:
            raise cg.CodegenError("cyclic non-centred reference through %r" % id_)
        z = L.q(offset[id_])
        if id_ in ncp_info:
            mu = resolve_scalar(L, ncp_info[id_]["mu"], stack + (id_,))
            sigma = resolve_scalar(L, ncp_info[id_]["sigma"], stack + (id_,))
            return L.g.add(mu, L.g.mul(sigma, z))
        return cg._apply_transform(L.g, nodes[id_]["transform"], z)

    def resolve_scalar(L, v, stack=(), elem=None):
        if isinstance(v, str):
            return resolve_ref(L, v, stack)
        a = np.asarray(v, dtype=np.float64)
        if a.ndim == 0:
            return L.datum(float(a))
        if a.ndim == 1 and elem is not None:
            return L.datum(float(a[elem]))
        raise cg.CodegenError("params are scalars, vectors or refs")

    units = []
    for id_ in sorted(nodes):
        n = nodes[id_]
        if n["op"] == "rv":
            if id_ not in offset:
                continue

            def prior(L, id_=id_, n=n):
                params = {k: resolve_scalar(L, v) for k, v in n["params"].items()}
                z = L.q(offset[id_])
                x = cg._apply_transform(L.g, n["transform"], z)
                t = cg._logpdf(L.g, n["dist"], x, params)
                if n["transform"] is not None:
                    t = L.g.add(t, cg._log_abs_det_jacobian(L.g, n["transform"], z))
                return t
            units.append(prior)
        else:
            tgt = nodes[n["target"]]
            if tgt["op"] != "rv" or tgt["transform"] is not None:
                raise cg.CodegenError("obs target not covered")
            val = n["value"]
            lens = [len(np.asarray(v)) for v in tgt["params"].values()
                    if not isinstance(v, str) and np.asarray(v).ndim == 1]
            nelem = val.shape[0] if val.ndim == 1 else (lens[0] if lens else 0)
            if val.ndim == 0 and lens:
                raise cg.CodegenError("scalar obs of a vector-valued target")
            for e in ([None] if nelem == 0 else range(nelem)):
                def obs(L, tgt=tgt, val=val, e=e):
                    params = {k: resolve_scalar(L, v, elem=e) for k, v in tgt["params"].items()}
                    x = L.datum(float(val) if val.ndim == 0 else float(val[e]))
                    return cg._logpdf(L.g, tgt["dist"], x, params)
                units.append(obs)
    return units


def plan(ir, ncp=True):
    """None when the model has no vectorisable family or does not fit 16 lanes."""
    nodes, ncp_info = cg._apply_ncp(ir, ncp)
    observed = cg._observed_targets(ir)
    free = sorted(i for i, n in nodes.items() if n["op"] == "rv" and i not in observed)
    D = len(free)
    if D > G or D < 2:
        return None
    offset = {id_: k for k, id_ in enumerate(free)}
    units = _units(nodes, ncp_info, offset)
    recs = []
    for u in units:
        r = _Rec()
        u(r)
        recs.append(r)
    groups = {}
    for k, r in enumerate(recs):
        groups.setdefault(tuple(r.g.ops), []).append(k)
    families, scalar_units = [], []
    for sig, members in groups.items():
        # more units than lanes (a long observation vector): 16 at a time
        for lo in range(0, len(members), G):
            part = members[lo:lo + G]
            fam = _family(part, recs, D) if len(part) >= 2 else None
            if fam is None:
                scalar_units.extend(part)
            else:
                families.append(fam)
    if not families:
        return None
    families.sort(key=lambda f: f["members"][0])
    scalar_units.sort()
    return dict(D=D, free=free, ncp_info=ncp_info, nodes=nodes, units=units, recs=recs,
                families=families, scalar_units=scalar_units)


def _family(members, recs, D):
    if len(members) > G:
        return None
    r0 = recs[members[0]]
    qplan, private_pos = [], []
    for p in range(len(r0.qs)):
        col = [recs[m].qs[p] for m in members]
        if len(set(col)) == 1:
            qplan.append(("shared", col[0]))
        elif len(set(col)) == len(col):
            qplan.append(("private", None))
            private_pos.append(p)
        else:
            return None
    lane_of = {}
    if private_pos:
        for m in members:
            dims = {recs[m].qs[p] for p in private_pos}
            if len(dims) != 1:
                return None        # more than one private variable in a unit
            lane_of[m] = dims.pop()
        if len(set(lane_of.values())) != len(members):
            return None
    else:
        for k, m in enumerate(members):
            lane_of[m] = k
    dplan_kind = []
    for p in range(len(r0.ds)):
        col = [recs[m].ds[p] for m in members]
        dplan_kind.append("uniform" if len(set(col)) == 1 else "lane")
    return dict(members=members, qplan=qplan, lane_of=lane_of, dplan_kind=dplan_kind)


def generate(ir, ncp=True):
    """-> dict(text, vdata, ...) for the 16-lane layout, or None."""
    pl = plan(ir, ncp)
    if pl is None:
        return None
    D = pl["D"]
    g = _VGraph()
    units, recs = pl["units"], pl["recs"]
    # per-lane raw data table: one mask column per family, then its lane-varying constants
    ncol = 0
    table = []          # list of columns, each a list of G values
    vec_terms = []
    for fam in pl["families"]:
        mask_col = ncol
        ncol += 1
        mask = [0.0] * G
        for m in fam["members"]:
            mask[fam["lane_of"][m]] = 1.0
        table.append(mask)
        r0 = recs[fam["members"][0]]
        dplan = []
        for p, kind in enumerate(fam["dplan_kind"]):
            if kind == "uniform":
                dplan.append(("uniform", r0.ds[p]))
            else:
                col = [0.0] * G
                for m in fam["members"]:
                    col[fam["lane_of"][m]] = recs[m].ds[p]
                # lanes outside the family evaluate the template too (their result is dropped by a
                # select): give them the first unit's constants so that they stay finite
                first = recs[fam["members"][0]].ds[p]
                col = [c if mask[l] else first for l, c in enumerate(col)]
                dplan.append(("lane", ncol))
                table.append(col)
                ncol += 1
        t = units[fam["members"][0]](_Template(g, fam["qplan"], dplan))
        vec_terms.append(g.sel_gt(g.ldata(mask_col), g.lit(0.5), t, g.lit(0.0)))
    vec_lp = vec_terms[0]
    for t in vec_terms[1:]:
        vec_lp = g.add(vec_lp, t)
    scal_terms = [units[k](_Plain(g)) for k in pl["scalar_units"]]
    scal_lp = None
    for t in scal_terms:
        scal_lp = t if scal_lp is None else g.add(t, scal_lp)
    if scal_lp is None:
        scal_lp = g.lit(0.0)
    n_fwd = len(g.ops)
    av = cg._Grad(g, vec_lp)
    av.run(n_fwd)
    n_fwd2 = len(g.ops)
    if g.const[scal_lp]:
        s_adj = {}
    else:
        a_s = cg._Grad(g, scal_lp)
        a_s.run(n_fwd2)
        s_adj = a_s.adj
    qnode = [g.key.get(("q", i)) for i in range(D)]
    shared = [i for i in range(D) if qnode[i] is not None and av.adj.get(qnode[i]) is not None]
    s_out = [vec_lp] + [av.adj[qnode[i]] for i in shared]            # reduced over the group
    own = g.key.get(("qown",))
    gown = av.adj.get(own) if own is not None else None
    sg = [s_adj.get(qnode[i]) if qnode[i] is not None else None for i in range(D)]
    smap = [-1] * D
    for k, i in enumerate(shared):
        smap[i] = k + 1
    text, nvc, nlc, nvu = _emit(g, D, s_out, gown, sg, scal_lp, ncol)
    vdata = list(g.data)
    for l in range(G):
        vdata.extend(table[c][l] for c in range(ncol))
    qmask = 0
    for m in __import__("re").finditer(r"qs\[(\d+)\]", text):
        qmask |= 1 << int(m.group(1))
    macros = ["#define EXMC_GEN_VEC 1", "#define EXMC_GEN_NS %d" % len(s_out),
              "#define EXMC_GEN_QMASK %du   /* dimensions the lane function reads through qs[] */" % qmask,
              "#define EXMC_GEN_NVU %d" % nvu, "#define EXMC_GEN_NLR %d" % ncol,
              "#define EXMC_GEN_NVC %d" % nvc, "#define EXMC_GEN_NLC %d" % nlc,
              "#define EXMC_GEN_SMAP {%s}" % ", ".join(str(v) for v in smap)]
    return dict(text="\n".join(macros) + "\n\n" + text, vdata=np.asarray(vdata, dtype=np.float64),
                n_families=len(pl["families"]), n_scalar_units=len(pl["scalar_units"]),
                n_reduced=len(s_out))


def _emit(g, D, s_out, gown, sg, scal_lp, nlr):
    outputs = [x for x in s_out + [gown, scal_lp] + sg if x is not None]
    live, stack = set(), list(outputs)
    while stack:
        i = stack.pop()
        if i in live:
            continue
        live.add(i)
        if g.ops[i][0] not in ("lit", "data", "q", "qown", "ldata"):
            stack.extend(g.ops[i][1:])
    uslot, lslot = {}, {}     # uniform / lane constants read by dynamic code (or being outputs)

    def want(a):
        if g.const[a] and g.ops[a][0] != "lit":
            d = lslot if g.lane[a] else uslot
            if a not in d:
                d[a] = len(d)
    for a in outputs:
        want(a)
    for i in sorted(live):
        if g.const[i] or g.ops[i][0] in ("q", "qown"):
            continue
        for a in g.ops[i][1:]:
            want(a)
    # lane constants may read uniform constants: those need uniform slots too
    lhost, stack = set(), list(lslot)
    while stack:
        i = stack.pop()
        if i in lhost:
            continue
        lhost.add(i)
        if g.ops[i][0] not in ("lit", "data", "ldata"):
            for a in g.ops[i][1:]:
                if g.lane[a]:
                    stack.append(a)
                elif g.ops[a][0] != "lit":
                    want(a)
    uhost, stack = set(), list(uslot)
    while stack:
        i = stack.pop()
        if i in uhost:
            continue
        uhost.add(i)
        if g.ops[i][0] not in ("lit", "data"):
            stack.extend(g.ops[i][1:])

    def ref(i, where):
        op = g.ops[i]
        if op[0] == "lit":
            s = repr(float.fromhex(op[1]))
            return "(%s)" % s if s.startswith("-") else s
        if where == "dyn":
            if i in uslot:
                return "vc[%d]" % uslot[i]
            if i in lslot:
                return "lc[%d]" % lslot[i]
        if where == "lfold" and i in uslot:
            return "vc[%d]" % uslot[i]
        if op[0] == "data":
            return "vdata[%d]" % op[1]
        if op[0] == "ldata":
            return "raw[%d]" % op[1]
        if op[0] == "q":
            return "qs[%d]" % op[1]
        if op[0] == "qown":
            return "qown"
        return "t%d" % i

    def stmt(i, where):
        op = g.ops[i]
        a = [ref(x, where) for x in op[1:]]
        if op[0] in cg._BIN:
            e = "%s %s %s" % (a[0], cg._BIN[op[0]], a[1])
        elif op[0] == "neg":
            e = "-%s" % a[0]
        elif op[0] in cg._FN1:
            # the lane function holds a handful of transcendentals: they are inlined (EXMC_GENV_*),
            # unlike the one-lane body's tens of calls (EXMC_GEN_*, exmc_models.hpp)
            e = "%s(%s)" % (cg._FN1[op[0]].replace("EXMC_GEN_", "EXMC_GENV_"), a[0])
        elif op[0] in cg._FN2:
            e = "%s(%s, %s)" % (cg._FN2[op[0]], a[0], a[1])
        elif op[0] == "sel_gt":
            e = "(%s > %s) ? %s : %s" % tuple(a)
        else:
            raise cg.CodegenError("cannot emit %s" % op[0])
        return "  const double t%d = %s;" % (i, e)

    L = ["/* 16-lane layout (exmc_amd/codegen_vec.py): vdata = uniform constants, then 16 rows of"
         " EXMC_GEN_NLR\n * per-lane constants; vc / lc = what the host folds from them. */"]
    L.append("EXMC_GEN_HOST void exmc_gen_vfold(const double* vdata, double* vc) {")
    for i in sorted(uhost):
        if g.ops[i][0] not in ("lit", "data"):
            L.append(stmt(i, "ufold"))
    for i, k in sorted(uslot.items(), key=lambda kv: kv[1]):
        L.append("  vc[%d] = %s;" % (k, ref(i, "ufold")))
    L.append("  (void)vdata; (void)vc;")
    L.append("}")
    L.append("")
    L.append("EXMC_GEN_HOST void exmc_gen_vfold_lane(const double* vc, const double* raw, double* lc) {")
    for i in sorted(lhost):
        if g.ops[i][0] not in ("lit", "data", "ldata") and g.lane[i]:
            L.append(stmt(i, "lfold"))
    for i, k in sorted(lslot.items(), key=lambda kv: kv[1]):
        L.append("  lc[%d] = %s;" % (k, ref(i, "lfold")))
    L.append("  (void)vc; (void)raw; (void)lc;")
    L.append("}")
    L.append("")
    L.append("/* one lane: s[0] = partial log-density of the vectorised terms, s[1..] = partial adjoints of"
             "\n * the shared variables, *gown = adjoint of the lane's own variable, sg[] / *slp = the"
             " scalar part */")
    # everything above is included once; the lane function is a section of its own, so that a device
    # build can compile the same text a second time under another name with watched main-path
    # exp / log (exmc_models.hpp "fast window": EXMC_GEN_VEC_SECTION, EXMC_GENV_NAME, EXMC_GENV_CTX_DECL)
    L.append("#endif   /* !EXMC_GEN_VEC_SECTION */")
    L.append("#ifndef EXMC_GENV_NAME")
    L.append("#define EXMC_GENV_NAME exmc_gen_lane")
    L.append("#define EXMC_GENV_CTX_DECL")
    L.append("#endif")
    L.append("EXMC_GEN_FN void EXMC_GENV_NAME(const double* vc, const double* lc, const double* qs, "
             "double qown,\n                               double* s, double* gown, double* sg, double* slp"
             " EXMC_GENV_CTX_DECL) {")
    for i in sorted(live):
        if not g.const[i] and g.ops[i][0] not in ("q", "qown"):
            L.append(stmt(i, "dyn"))
    for k, x in enumerate(s_out):
        L.append("  s[%d] = %s;" % (k, ref(x, "dyn")))
    L.append("  *gown = %s;" % ("0.0" if gown is None else ref(gown, "dyn")))
    for i in range(D):
        L.append("  sg[%d] = %s;" % (i, "0.0" if sg[i] is None else ref(sg[i], "dyn")))
    L.append("  *slp = %s;" % ref(scal_lp, "dyn"))
    L.append("  (void)vc; (void)lc; (void)qs; (void)qown;")
    L.append("}")
    return "\n".join(L) + "\n", max(1, len(uslot)), max(1, len(lslot)), len(g.data)
