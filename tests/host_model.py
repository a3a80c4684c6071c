"""Test infrastructure: the oracle behind the method names of ``model.ExaModel`` (torch CPU tensors), and a dense host
factorisation behind the linear-system interface of ``ipm.InteriorPointSolver`` — so that the SOLVER LOGIC (barrier updates,
line search, inertia correction) is exercised in the CPU suite.  The product path uses the device model and the chain KKT
solver; nothing in the package imports this file."""
import numpy as np
import torch

from pyoracle import OracleModel


class HostModel:
    def __init__(self, blob: bytes):
        self.om = OracleModel(blob)
        self.meta = self.om               # nvar, ncon, nnzj, nnzh, x0, lvar, uvar, lcon, ucon, minimize
        self.device = torch.device("cpu")
        self.jr, self.jc = self.om.jac_structure(base=0)
        self.hr, self.hc = self.om.hess_structure(base=0)

    @staticmethod
    def _np(t):
        return t.detach().numpy() if isinstance(t, torch.Tensor) else np.asarray(t, dtype=np.float64)

    def obj(self, x):
        return float(self.om.obj(self._np(x)))

    def grad(self, x, g=None):
        v = torch.from_numpy(np.ascontiguousarray(self.om.grad(self._np(x))))
        return v if g is None else g.copy_(v)

    def cons(self, x, c=None):
        v = torch.from_numpy(np.ascontiguousarray(self.om.cons(self._np(x))))
        return v if c is None else c.copy_(v)

    def jtprod(self, x, v, out=None):
        w = torch.from_numpy(np.ascontiguousarray(self.om.jtprod(self._np(x), self._np(v))))
        return w if out is None else out.copy_(w)

    def jac_structure_device(self, base=0):
        return torch.from_numpy(self.jr + base), torch.from_numpy(self.jc + base)

    def jac_hess_coord(self, x, y, jac=None, hess=None, obj_weight=1.0):
        jac.copy_(torch.from_numpy(self.om.jac_coord(self._np(x))))
        hess.copy_(torch.from_numpy(self.om.hess_coord(self._np(x), self._np(y), float(obj_weight))))
        return jac, hess


class HostLinear:
    """K = [H + diag(sigma) + dw I, J'; J, -diag(dc)] dense; inertia from its LDL' factorisation"""

    def __init__(self, model: HostModel):
        self.m = model
        self.n, self.mc = model.om.nvar, model.om.ncon

    def assemble(self, hv, jv, sigma, dw, dc):
        n, mc, M = self.n, self.mc, self.m
        K = np.zeros((n + mc, n + mc))
        h = hv.numpy()
        np.add.at(K, (M.hr, M.hc), h)
        off = M.hr != M.hc
        np.add.at(K, (M.hc[off], M.hr[off]), h[off])
        j = jv.numpy()
        np.add.at(K, (n + M.jr, M.jc), j)
        np.add.at(K, (M.jc, n + M.jr), j)
        K[np.arange(n), np.arange(n)] += (0.0 if sigma is None else sigma.numpy()) + dw
        K[n + np.arange(mc), n + np.arange(mc)] -= dc.numpy() if isinstance(dc, torch.Tensor) else dc
        self.K = K

    def factor(self):
        # Bunch-Kaufman LDL': the inertia of K is that of the block-diagonal factor (eigenvalues of K itself lose the
        # -delta_c pivots next to barrier terms of 1e10)
        from scipy.linalg import ldl
        _, d, _ = ldl(self.K)
        neg = zero = 0
        i, N = 0, d.shape[0]
        while i < N:
            if i + 1 < N and d[i + 1, i] != 0.0:
                ev = np.linalg.eigvalsh(d[i:i + 2, i:i + 2])
                i += 2
            else:
                ev = np.array([d[i, i]])
                i += 1
            neg += int((ev < 0).sum())
            zero += int((ev == 0).sum())
        self._inertia = (N - neg - zero, neg, zero)

    def inertia(self):
        return self._inertia

    def solve(self, rhs, refine="auto", rtol=1e-8):
        return torch.from_numpy(np.linalg.solve(self.K, rhs.numpy()))
