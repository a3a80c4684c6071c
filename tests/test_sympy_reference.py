"""Third opinion on the derivatives: symbolic differentiation.

For small models every template item is turned into a sympy expression over symbols x_i (walking
the *Python* trees, like helpers.TorchModel), differentiated symbolically, and evaluated with
30-digit mpmath arithmetic.  The oracle's COO Jacobian / Lagrangian Hessian, scattered to dense,
must agree to a few ulps — this is independent of both the oracle's reverse sweeps and torch's
autograd, and it states the oracle's actual rounding error against exact derivatives."""
import numpy as np
import pytest
import mpmath
import sympy as sp

import cases
from helpers import coo_to_dense, lower_to_full
from infiniteexamodels.jl_amd import nodes as N
from infiniteexamodels.jl_amd.core import T_CON, T_OBJ
from pyoracle import OracleModel

D2R = sp.pi / 180
UN = {
    "neg": lambda a: -a, "pos": lambda a: a, "inv": lambda a: 1 / a, "sqrt": sp.sqrt, "cbrt": lambda a: a / (a * a) ** sp.Rational(1, 3),
    "abs": lambda a: sp.sqrt(a * a), "abs2": lambda a: a * a, "exp": sp.exp, "exp2": lambda a: 2 ** a, "log": sp.log,
    "log2": lambda a: sp.log(a) / sp.log(2), "log10": lambda a: sp.log(a) / sp.log(10), "log1p": lambda a: sp.log(1 + a),
    "sin": sp.sin, "cos": sp.cos, "tan": sp.tan, "asin": sp.asin, "acos": sp.acos, "csc": lambda a: 1 / sp.sin(a),
    "sec": lambda a: 1 / sp.cos(a), "cot": lambda a: 1 / sp.tan(a), "atan": sp.atan, "acot": lambda a: sp.atan(1 / a),
    "sind": lambda a: sp.sin(a * D2R), "cosd": lambda a: sp.cos(a * D2R), "tand": lambda a: sp.tan(a * D2R),
    "cscd": lambda a: 1 / sp.sin(a * D2R), "secd": lambda a: 1 / sp.cos(a * D2R), "cotd": lambda a: 1 / sp.tan(a * D2R),
    "atand": lambda a: sp.atan(a) / D2R, "acotd": lambda a: sp.atan(1 / a) / D2R, "sinh": sp.sinh, "cosh": sp.cosh,
    "tanh": sp.tanh, "csch": lambda a: 1 / sp.sinh(a), "sech": lambda a: 1 / sp.cosh(a), "coth": lambda a: 1 / sp.tanh(a),
    "atanh": sp.atanh, "acoth": lambda a: sp.atanh(1 / a),
}


def exact(v):
    return sp.Float(float(v), 40)      # the binary64 value, exactly


class SympyModel:
    """One symbolic derivation per TEMPLATE (a symbol per distinct leaf), evaluated per item."""

    def __init__(self, core):
        self.core = core

    def index(self, items, k, key):
        c0, terms = key
        return c0 + sum(coef * int(items.column(name)[k]) for name, coef in terms) - 1

    def expr(self, node, leaves):
        def leaf(kind, key):
            return leaves.setdefault((kind, key), sp.Symbol(f"{kind}{len(leaves)}", real=True))
        if isinstance(node, (N.Null, N.Const)):
            return exact(node.value)
        if isinstance(node, N.DataField):
            return leaf("d", node.name)
        if isinstance(node, (N.Var, N.ParameterNode)):
            c0, terms = N.affine_index(node.i) if isinstance(node.i, N.Node) else (int(node.i), {})
            return leaf("v" if isinstance(node, N.Var) else "p", (c0, tuple(sorted(terms.items()))))
        if isinstance(node, N.Unary):
            return UN[node.op](self.expr(node.inner, leaves))
        a, b = self.expr(node.inner1, leaves), self.expr(node.inner2, leaves)
        return {"+": lambda: a + b, "-": lambda: a - b, "*": lambda: a * b, "/": lambda: a / b, "^": lambda: a ** b}[node.op]()

    def dense(self, x, y, w):
        n = self.core.nvar
        g, J, H, c, f, row = np.zeros(n), [], np.zeros((n, n)), [], 0.0, 0
        mods = [{"DiracDelta": lambda *a: mpmath.mpf(0)}, "mpmath"]
        for t in self.core.templates:
            leaves = {}
            e = self.expr(t.expr, leaves)
            keys = list(leaves)
            syms = [leaves[k] for k in keys]
            vs = [i for i, k in enumerate(keys) if k[0] == "v"]
            d1 = [sp.diff(e, syms[i]) for i in vs]
            d2 = [[sp.diff(d, syms[j]) for j in vs] for d in d1]
            fn = sp.lambdify(syms, [e] + d1 + [q for r in d2 for q in r], mods)
            for k in range(len(t.items)):
                pt = []
                for kind, key in keys:
                    if kind == "d":
                        pt.append(mpmath.mpf(float(t.items.column(key)[k])))
                    elif kind == "p":
                        pt.append(mpmath.mpf(float(self.core.theta[self.index(t.items, k, key)])))
                    else:
                        pt.append(mpmath.mpf(float(x[self.index(t.items, k, key)])))
                out = [float(v) for v in fn(*pt)]
                ids = [self.index(t.items, k, keys[i][1]) for i in vs]
                nv = len(vs)
                if t.kind == T_OBJ:
                    f += out[0]
                    scale = w
                    for i, v in zip(ids, out[1:1 + nv]):
                        g[i] += v
                else:
                    c.append(out[0])
                    r = np.zeros(n)
                    for i, v in zip(ids, out[1:1 + nv]):
                        r[i] += v
                    J.append(r)
                    scale = y[row]
                    row += 1
                for a in range(nv):
                    for b in range(nv):
                        H[ids[a], ids[b]] += scale * out[1 + nv + a * nv + b]
        return f, np.array(c), g, np.array(J).reshape(len(c), n), H


@pytest.mark.parametrize("name", ["quadrotor_5", "rosenbrock", "pfun", "operator_zoo", "opf_7", "farmer_5", "quadrotor_oc3_40", "pandemic_20x3", "irregular",
                                  "hovercraft_oc4", "three_node_50", "kinetic_20", "test_problem_1_oc3",
                                  "test_problem_2_obj1", "test_problem_2_obj3", "test_problem_2_obj4", "pfun_full"])
def test_oracle_matches_symbolic_derivatives(name):
    core = cases.build_core(name)
    om = OracleModel(core.to_blob())
    x, y = cases.eval_point_for(name, om, seed=3)
    w = 0.8
    with mpmath.workdps(30):
        f, c, g, J, H = SympyModel(core).dense(x, y, w)
    tol = lambda ref: 1e-13 * max(1.0, float(np.abs(ref).max()) if ref.size else 1.0)
    assert abs(om.obj(x) - f) <= 1e-13 * max(1.0, abs(f))
    np.testing.assert_allclose(om.cons(x), c, rtol=0, atol=tol(c))
    np.testing.assert_allclose(om.grad(x), g, rtol=0, atol=tol(g))
    jr, jc = om.jac_structure()
    np.testing.assert_allclose(coo_to_dense(jr, jc, om.jac_coord(x), J.shape), J, rtol=0, atol=tol(J))
    hr, hc = om.hess_structure()
    Ho = lower_to_full(coo_to_dense(hr, hc, om.hess_coord(x, y, w), H.shape))
    np.testing.assert_allclose(Ho, H, rtol=0, atol=tol(H))
