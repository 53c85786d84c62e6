"""TEST INFRASTRUCTURE. A model object for tests/py_tree.py / py_sampler.py whose leapfrog steps come out of a
fused-chain provider the way the reference's speculative path takes them (lib/exmc/nuts/tree.ex:509-612 the
buffer, :613-653 the dispatch): a direction's next K = 32 steps are asked for in ONE call of
`leapfrog_chain_normal(q, p, inv_mass, k, signed_eps, mu, sigma)` and served row by row; a leaf that starts
anywhere else than at the last row served starts a new chain. What the reference's own test of the hook checks
(test/nuts/fused_chain_diag_test.exs: x ~ N(0, 1), seed 42, 200 + 1000, var in [0.7, 1.3]) is then run with
`chain_fn` = the checker's statement (CPU suite) or the HIP entry point (GPU suite)."""
import ctypes as C

import numpy as np

import oracle as O

K = 32          # tree.ex:515 max(32, 2 * need); the notebook's "K=32 leapfrog steps in a single GPU dispatch"


class FusedChainModel:
    def __init__(self, d, mu, sigma, chain_fn, cfg=None):
        assert mu == 0.0 and sigma == 1.0, "logp_grad below is the checker's N(0, 1) kind"
        self.d, self.mu, self.sigma, self.chain_fn = d, mu, sigma, chain_fn
        self.base = O.std_normal(d)
        self.cfg = cfg or O.Cfg(0, 1)
        self.buf = {}           # direction sign -> (rows, cursor, eps)
        self.dispatches = 0
        self.steps = 0

    def logp_grad(self, q, cfg=None):
        return self.base.logp_grad(q, cfg or self.cfg)

    def leapfrog(self, q, p, g, eps, inv_mass):
        """One leaf: (q', p', logp', g', joint logp') -- py_tree.Tree.leaf's provider."""
        sign = 1 if eps > 0 else -1
        q, p = np.asarray(q, dtype=np.float64), np.asarray(p, dtype=np.float64)
        b = self.buf.get(sign)
        if b is not None:
            rows, cur, beps = b
            ok = (beps == eps and cur < K and np.array_equal(rows[0][cur - 1], q)
                  and np.array_equal(rows[1][cur - 1], p))
            if not ok:
                b = None
        if b is None:
            rows = self.chain_fn(q, p, inv_mass, K, eps, self.mu, self.sigma)      # (all_q, all_p, all_logp, all_grad)
            self.dispatches += 1
            cur = 0
        self.buf[sign] = (rows, cur + 1, eps)
        self.steps += 1
        qn, pn, lp, gn = rows[0][cur].copy(), rows[1][cur].copy(), float(rows[2][cur]), rows[3][cur].copy()
        im = np.ascontiguousarray(np.asarray(inv_mass, dtype=np.float64))
        ke = O.lib().exo_kinetic_energy(O.dptr(np.ascontiguousarray(pn)), O.dptr(im), self.d, self.cfg)
        return qn, pn, lp, gn, lp - ke
