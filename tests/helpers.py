"""Shared test helpers: an evaluator that is independent of the oracle.

``TorchModel`` walks the *Python* expression trees recorded by ``ExaCore`` (not the
blob, not the C code) with torch float64 tensors, vectorised over items, and lets
torch autograd produce dense gradients / Jacobians / Hessians.  The oracle's COO
output, scattered to dense, must agree with it.
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "oracle") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))

from infiniteexamodels.jl_amd import nodes as N  # noqa: E402
from infiniteexamodels.jl_amd.core import T_CON, T_OBJ  # noqa: E402

D2R = math.pi / 180.0

_UN = {
    "neg": lambda x: -x, "pos": lambda x: x, "inv": lambda x: 1.0 / x, "sqrt": torch.sqrt,
    "cbrt": lambda x: torch.sign(x) * torch.abs(x) ** (1.0 / 3.0), "abs": torch.abs,
    "abs2": lambda x: x * x, "exp": torch.exp, "exp2": torch.exp2, "log": torch.log,
    "log2": torch.log2, "log10": torch.log10, "log1p": torch.log1p,
    "sin": torch.sin, "cos": torch.cos, "tan": torch.tan, "asin": torch.asin, "acos": torch.acos,
    "csc": lambda x: 1.0 / torch.sin(x), "sec": lambda x: 1.0 / torch.cos(x),
    "cot": lambda x: 1.0 / torch.tan(x), "atan": torch.atan, "acot": lambda x: torch.atan(1.0 / x),
    "sind": lambda x: torch.sin(x * D2R), "cosd": lambda x: torch.cos(x * D2R),
    "tand": lambda x: torch.tan(x * D2R), "cscd": lambda x: 1.0 / torch.sin(x * D2R),
    "secd": lambda x: 1.0 / torch.cos(x * D2R), "cotd": lambda x: 1.0 / torch.tan(x * D2R),
    "atand": lambda x: torch.atan(x) / D2R, "acotd": lambda x: torch.atan(1.0 / x) / D2R,
    "sinh": torch.sinh, "cosh": torch.cosh, "tanh": torch.tanh,
    "csch": lambda x: 1.0 / torch.sinh(x), "sech": lambda x: 1.0 / torch.cosh(x),
    "coth": lambda x: 1.0 / torch.tanh(x), "atanh": torch.atanh,
    "acoth": lambda x: torch.atanh(1.0 / x),
}


class TorchModel:
    def __init__(self, core):
        self.core = core
        self.theta = torch.tensor(core.theta, dtype=torch.float64)

    def _index(self, items, i):
        c0, terms = N.affine_index(i) if isinstance(i, N.Node) else (int(i), {})
        idx = np.full(len(items), c0, dtype=np.int64)
        for name, coef in terms.items():
            idx = idx + coef * items.column(name).astype(np.int64)
        return torch.from_numpy(idx - 1)

    def _eval(self, node, items, x):
        if isinstance(node, N.Null):
            return torch.full((len(items),), node.value, dtype=torch.float64)
        if isinstance(node, N.Const):
            return torch.full((len(items),), node.value, dtype=torch.float64)
        if isinstance(node, N.DataField):
            return torch.from_numpy(items.column(node.name).astype(np.float64))
        if isinstance(node, N.Var):
            return x[self._index(items, node.i)]
        if isinstance(node, N.ParameterNode):
            return self.theta[self._index(items, node.i)]
        if isinstance(node, N.Unary):
            return _UN[node.op](self._eval(node.inner, items, x))
        a = self._eval(node.inner1, items, x)
        b = self._eval(node.inner2, items, x)
        return {"+": a + b, "-": a - b, "*": a * b, "/": a / b}.get(node.op) if node.op != "^" else a ** b

    def obj(self, x):
        tot = torch.zeros((), dtype=torch.float64)
        for t in self.core.templates:
            if t.kind == T_OBJ:
                tot = tot + self._eval(t.expr, t.items, x).sum()
        return tot

    def cons(self, x):
        out = []
        for t in self.core.templates:
            if t.kind == T_CON:
                out.append(self._eval(t.expr, t.items, x))
        return torch.cat(out) if out else torch.zeros(0, dtype=torch.float64)

    def dense(self, x, y, obj_weight):
        """f, c, ∇f, J, ∇²L as dense numpy arrays."""
        xt = torch.tensor(np.asarray(x), dtype=torch.float64)
        yt = torch.tensor(np.asarray(y), dtype=torch.float64)
        f = self.obj(xt)
        c = self.cons(xt)
        g = torch.autograd.functional.jacobian(self.obj, xt)
        J = torch.autograd.functional.jacobian(self.cons, xt) if c.numel() else torch.zeros(0, xt.numel())
        lag = lambda z: obj_weight * self.obj(z) + (yt * self.cons(z)).sum()
        H = torch.autograd.functional.hessian(lag, xt)
        return f.item(), c.numpy(), g.numpy(), J.numpy(), H.numpy()


def coo_to_dense(rows, cols, vals, shape):
    M = np.zeros(shape)
    np.add.at(M, (rows, cols), vals)
    return M


def lower_to_full(L):
    return L + L.T - np.diag(np.diag(L))


def eval_point(model, seed=0, scale=0.1, clip=None):
    """Seeded evaluation point of SURVEY §8(d): x = x0 + scale·N(0,1), y ~ N(0,1)."""
    x = model.x0 + scale * np.random.default_rng(seed).standard_normal(model.nvar)
    if clip is not None:
        x = np.clip(x, -clip, clip)
    y = np.random.default_rng(seed + 1).standard_normal(model.ncon)
    return x, y
