"""A small InfiniteOpt-style modelling layer — just enough to state the reference's
workloads (``/root/reference/examples/*.jl``, ``ESCAPE34/*.jl``) and test problems
(``test/solve.jl``, ``test/ipopt.jl``, ``test/transcription.jl``) so that
:mod:`.transcribe` can mirror ``src/transform.jl`` on them.

Conventions taken from InfiniteOpt.jl 0.6 [EXT — not vendored; validated against the
reference's test constants, SURVEY.md Appendix A]:

* ``num_supports = N`` on ``[lb, ub]`` → ``range(lb, ub, length = N)``, rounded to 12
  significant digits; ``add_supports`` merges and sorts;
* ``∫(f, p)`` on a bounded scalar parameter → trapezoid over the parameter's supports;
  ``𝔼(f, ξ)`` → equal weights ``1/N``;
* default derivative method: backward finite difference.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from .jump_expr import (AffExpr, ConstraintSpec, NonlinearExpr, QuadExpr, Scalar, VariableRef,
                        all_expression_variables, is_number, nl)

# nonlinear operator functions for model expressions: sin(x), cos(x), …
from .nodes import UNARY_OPS as _UNARY

for _name in _UNARY:
    if _name not in ("neg", "pos"):
        globals()[_name] = nl(_name)


def OrthogonalCollocation(num_nodes: int):
    """``OrthogonalCollocation(num_nodes)`` (Gauss–Lobatto; ``num_nodes`` counts the two element
    boundaries, so ``num_nodes - 2`` internal supports are generated per interval)."""
    if num_nodes < 2:
        raise ValueError("OrthogonalCollocation needs at least 2 nodes")
    return ("oc", int(num_nodes))


def FiniteDifference(kind: str = "backward"):
    return ("fd_" + kind,)


def lobatto_internal_nodes(num_nodes: int) -> np.ndarray:
    """Interior Gauss–Lobatto nodes on (−1, 1): the roots of P'_{n−1} (n = num_nodes)."""
    n = num_nodes - 1
    if n < 2:
        return np.zeros(0)
    c = np.zeros(n + 1)
    c[n] = 1.0
    return np.sort(np.polynomial.legendre.legroots(np.polynomial.legendre.legder(c)))


def _round_sig(a: np.ndarray, sig: int = 12) -> np.ndarray:
    a = np.asarray(a, dtype=np.float64)
    out = a.copy()
    nz = a != 0
    mag = np.floor(np.log10(np.abs(a[nz])))
    scale = 10.0 ** (sig - 1 - mag)
    out[nz] = np.round(a[nz] * scale) / scale
    return out


# ---------------------------------------------------------------------------
# reference kinds
# ---------------------------------------------------------------------------
class ParameterGroup:
    """One entry of ``InfiniteOpt.parameter_refs(model)``: an independent scalar
    parameter or a vector of dependent parameters sharing supports."""

    def __init__(self, index: int, dependent: bool):
        self.index = index          # 1-based group index
        self.dependent = dependent
        self.prefs: List["InfiniteParameterRef"] = []
        self.supports = np.zeros((0, 0))   # (n_supports, n_prefs); sorted for independent
        self.derivative_method = ("fd_backward",)
        self.lb = self.ub = None
        self.public_supports = None      # set once generative (collocation) supports are merged in
        self.internal = None             # bool per support: generated internal collocation node

    @property
    def num_supports(self) -> int:
        return self.supports.shape[0]


class InfiniteParameterRef(VariableRef):
    def __init__(self, model, name, group: ParameterGroup, pos: int):
        super().__init__(model, "infinite_parameter", name)
        self.group, self.pos = group, pos

    @property
    def supports(self) -> np.ndarray:
        return self.group.supports[:, self.pos]


class FiniteParameterRef(VariableRef):
    def __init__(self, model, name, value):
        super().__init__(model, "finite_parameter", name)
        self.value = float(value)


class ParameterFunctionRef(VariableRef):
    def __init__(self, model, name, func, prefs):
        super().__init__(model, "parameter_function", name)
        self.func, self.prefs = func, list(prefs)

    @property
    def group_idxs(self) -> List[int]:
        return _group_idxs(self.prefs)

    def __call__(self, *args):
        """``pf(0.5, s)`` — a parameter function restricted to some parameter values."""
        if len(args) != len(self.prefs):
            raise ValueError("wrong number of arguments")
        if all(is_number(a) for a in args):
            return float(self.func(*args))
        return self.model._semi(self, list(args))


class VarInfo:
    def __init__(self, lb=None, ub=None, fix=None, start=None):
        self.lb, self.ub, self.fix, self.start = lb, ub, fix, start


class FiniteVariableRef(VariableRef):
    def __init__(self, model, name, info: VarInfo):
        super().__init__(model, "finite_variable", name)
        self.info = info


class InfiniteVariableRef(VariableRef):
    """Infinite variable or derivative variable (``kind`` tells which)."""

    def __init__(self, model, name, prefs, info: VarInfo, kind="infinite_variable"):
        super().__init__(model, kind, name)
        self.prefs, self.info = list(prefs), info

    @property
    def group_idxs(self) -> List[int]:
        return _group_idxs(self.prefs)

    def __call__(self, *args):
        """``y(0, x)`` → semi-infinite variable, ``y(0, 1)`` → point variable."""
        if len(args) != len(self.prefs):
            raise ValueError("wrong number of arguments")
        if all(is_number(a) for a in args):
            return self.model._point(self, [float(a) for a in args])
        for a, p in zip(args, self.prefs):
            if not is_number(a) and a is not p:
                raise ValueError("arguments must be numbers or the variable's own parameters")
        return self.model._semi(self, list(args))


class DerivativeRef(InfiniteVariableRef):
    def __init__(self, model, name, arg, pref, info):
        prefs = arg.prefs if isinstance(arg, InfiniteVariableRef) else arg.free_prefs
        super().__init__(model, name, prefs, info, kind="derivative")
        self.arg, self.pref, self.order = arg, pref, 1


class SemiInfiniteVariableRef(VariableRef):
    def __init__(self, model, name, ivref: InfiniteVariableRef, args):
        super().__init__(model, "semi_infinite_variable", name)
        self.ivref, self.args = ivref, args   # args: number (fixed) or pref (free)
        self.info = VarInfo()

    @property
    def free_prefs(self):
        return [a for a in self.args if not is_number(a)]

    @property
    def prefs(self):
        return self.free_prefs

    @property
    def group_idxs(self) -> List[int]:
        return _group_idxs(self.free_prefs)


class PointVariableRef(VariableRef):
    def __init__(self, model, name, ivref, values):
        super().__init__(model, "point_variable", name)
        self.ivref, self.values = ivref, values
        self.info = VarInfo()


class MeasureRef(VariableRef):
    def __init__(self, model, name, func, prefs, supports, coeffs):
        super().__init__(model, "measure", name)
        self.func = func
        self.prefs = prefs            # list of parameter refs (1 for ∫, whole group for 𝔼)
        self.supports = supports      # (n, len(prefs))
        self.coeffs = coeffs


def _group_idxs(prefs) -> List[int]:
    out = []
    for p in prefs:
        if p.group.index not in out:
            out.append(p.group.index)
    return out


def parameter_group_int_indices(expr) -> List[int]:
    """Sorted group indices an expression / reference depends on
    (``InfiniteOpt.parameter_group_int_indices``)."""
    groups = set()

    def visit(v):
        if isinstance(v, InfiniteParameterRef):
            groups.add(v.group.index)
        elif isinstance(v, (InfiniteVariableRef, SemiInfiniteVariableRef, ParameterFunctionRef)):
            groups.update(v.group_idxs)
        elif isinstance(v, MeasureRef):
            inner = set(parameter_group_int_indices(v.func))
            inner -= {p.group.index for p in v.prefs}
            groups.update(inner)

    for v in ([expr] if isinstance(expr, VariableRef) else all_expression_variables(expr)):
        visit(v)
    return sorted(groups)


class DomainRestriction:
    """``DomainRestriction(f, prefs…)``: keep supports where ``f(values…)`` is true."""

    def __init__(self, func: Callable, *prefs):
        self.func, self.parameter_refs = func, list(prefs)

    def __call__(self, supp: Sequence[float]) -> bool:
        return bool(self.func(*supp))


class ConstraintObject:
    def __init__(self, func, lb: float, ub: float, restriction: Optional[DomainRestriction], name=""):
        self.func, self.lb, self.ub, self.restriction, self.name = func, lb, ub, restriction, name
        self.mapping = None


# ---------------------------------------------------------------------------
# the model
# ---------------------------------------------------------------------------
class InfiniteModel:
    """``InfiniteOpt.InfiniteModel([backend])``."""

    def __init__(self, backend=None):
        self.groups: List[ParameterGroup] = []
        self.finite_parameters: List[FiniteParameterRef] = []
        self.parameter_functions: List[ParameterFunctionRef] = []
        self.finite_variables: List[FiniteVariableRef] = []
        self.infinite_variables: List[InfiniteVariableRef] = []
        self.derivatives: List[DerivativeRef] = []
        self.semi_infinite_variables: List[SemiInfiniteVariableRef] = []
        self.point_variables: List[PointVariableRef] = []
        self.constraints: List[ConstraintObject] = []
        self.piecewise_vars: Dict[InfiniteParameterRef, list] = {}
        self.objective_sense: Optional[str] = None
        self.objective_function = None
        self.backend = backend
        self._ready = False
        if backend is not None and hasattr(backend, "_attach"):
            backend._attach(self)

    # -- parameters ---------------------------------------------------------
    def infinite_parameter(self, name: str, lb: float = None, ub: float = None, num_supports: int = 0,
                           supports: Sequence[float] = None, derivative_method=None) -> InfiniteParameterRef:
        g = ParameterGroup(len(self.groups) + 1, dependent=False)
        g.lb, g.ub = lb, ub
        if derivative_method is not None:
            g.derivative_method = tuple(derivative_method)
        if supports is not None:
            s = np.unique(_round_sig(np.asarray(supports, dtype=np.float64)))
        else:
            s = _round_sig(np.linspace(lb, ub, num_supports)) if num_supports else np.zeros(0)
        g.supports = s.reshape(-1, 1)
        p = InfiniteParameterRef(self, name, g, 0)
        g.prefs.append(p)
        self.groups.append(g)
        self._ready = False
        return p

    def dependent_parameters(self, names: Sequence[str], supports: np.ndarray) -> List[InfiniteParameterRef]:
        """``@infinite_parameter(m, ξ[1:n] ~ dist, num_supports = N)`` with the sampled
        supports given explicitly (``supports[k, c]``; Julia's RNG stream is not reproduced)."""
        g = ParameterGroup(len(self.groups) + 1, dependent=True)
        g.supports = _round_sig(np.asarray(supports, dtype=np.float64).reshape(len(supports), len(names)))
        for c, n in enumerate(names):
            g.prefs.append(InfiniteParameterRef(self, n, g, c))
        self.groups.append(g)
        self._ready = False
        return list(g.prefs)

    def add_supports(self, pref: InfiniteParameterRef, values: Sequence[float]):
        g = pref.group
        assert not g.dependent
        s = np.concatenate([g.supports[:, 0], _round_sig(np.asarray(values, dtype=np.float64))])
        g.supports = np.unique(s).reshape(-1, 1)
        self._ready = False

    def set_derivative_method(self, pref: InfiniteParameterRef, method) -> None:
        pref.group.derivative_method = tuple(method)
        self._ready = False

    def constant_over_collocation(self, var, pref: InfiniteParameterRef) -> None:
        """``constant_over_collocation(u, t)``: ``u`` is held constant over the internal
        collocation supports of each element (transform.jl:565-601)."""
        self.piecewise_vars.setdefault(pref, [])
        if all(v is not var for v in self.piecewise_vars[pref]):
            self.piecewise_vars[pref].append(var)
        self._ready = False

    def add_generative_supports(self, g: ParameterGroup) -> None:
        """``InfiniteOpt.add_generative_supports``: merge the internal collocation nodes of every
        interval into the group's supports (idempotent)."""
        if g.derivative_method[0] != "oc" or g.dependent:
            return
        if g.public_supports is None:
            g.public_supports = g.supports[:, 0].copy()
        pub = g.public_supports
        nodes = lobatto_internal_nodes(g.derivative_method[1])
        if len(nodes) == 0 or len(pub) < 2:
            g.supports = pub.reshape(-1, 1)
            g.internal = np.zeros(len(pub), dtype=bool)
            return
        lo, hi = pub[:-1], pub[1:]
        inner = (lo[:, None] + hi[:, None]) / 2 + (hi[:, None] - lo[:, None]) / 2 * nodes[None, :]
        allp = np.concatenate([np.column_stack([lo, _round_sig(inner)]).reshape(-1), pub[-1:]])
        flag = np.concatenate([np.tile(np.r_[False, np.ones(len(nodes), dtype=bool)], len(lo)), [False]])
        g.supports = allp.reshape(-1, 1)
        g.internal = flag

    def finite_parameter(self, name: str, value: float) -> FiniteParameterRef:
        p = FiniteParameterRef(self, name, value)
        self.finite_parameters.append(p)
        self._ready = False
        return p

    def parameter_function(self, name: str, func: Callable, *prefs) -> ParameterFunctionRef:
        pf = ParameterFunctionRef(self, name, func, prefs)
        self.parameter_functions.append(pf)
        self._ready = False
        return pf

    # -- variables ------------------------------------------------------------
    def variable(self, name: str, *prefs, lb=None, ub=None, fix=None, start=None):
        info = VarInfo(lb, ub, fix, start)
        if prefs:
            v = InfiniteVariableRef(self, name, prefs, info)
            self.infinite_variables.append(v)
        else:
            v = FiniteVariableRef(self, name, info)
            self.finite_variables.append(v)
        self._ready = False
        return v

    def deriv(self, arg, pref: InfiniteParameterRef) -> DerivativeRef:
        """``∂(y, t)`` (first order; repeated calls return the same derivative)."""
        for d in self.derivatives:
            if d.arg is arg and d.pref is pref:
                return d
        d = DerivativeRef(self, f"∂({arg!r},{pref!r})", arg, pref, VarInfo())
        self.derivatives.append(d)
        self._ready = False
        return d

    def _semi(self, ivref, args) -> SemiInfiniteVariableRef:
        for s in self.semi_infinite_variables:
            if s.ivref is ivref and len(s.args) == len(args) and all(
                    (a is b) if not is_number(a) else (is_number(b) and a == b) for a, b in zip(s.args, args)):
                return s
        s = SemiInfiniteVariableRef(self, f"{ivref!r}{tuple(args)!r}", ivref, args)
        self.semi_infinite_variables.append(s)
        self._ready = False
        return s

    def _point(self, ivref, values) -> PointVariableRef:
        for p in self.point_variables:
            if p.ivref is ivref and p.values == values:
                return p
        p = PointVariableRef(self, f"{ivref!r}{tuple(values)!r}", ivref, values)
        self.point_variables.append(p)
        self._ready = False
        return p

    # -- measures ---------------------------------------------------------------
    def integral(self, func, pref: InfiniteParameterRef) -> MeasureRef:
        """``∫(func, pref)`` — trapezoid over the supports present at build time."""
        m = MeasureRef(self, f"∫({pref!r})", func, [pref], None, None)
        m.method = "trapezoid"
        return m

    def expect(self, func, pref_or_group) -> MeasureRef:
        """``𝔼(func, ξ)`` — equal-weight sample average over the group's supports."""
        prefs = list(pref_or_group) if isinstance(pref_or_group, (list, tuple)) else [pref_or_group]
        m = MeasureRef(self, "𝔼", func, prefs, None, None)
        m.method = "expect"
        return m

    # -- constraints & objective --------------------------------------------------
    def constraint(self, spec: ConstraintSpec, restriction: DomainRestriction = None, name: str = "",
                   lower: float = None, upper: float = None) -> ConstraintObject:
        f = spec.func
        off = 0.0
        if isinstance(f, AffExpr):
            off, f = f.constant, AffExpr(OrderedDict(f.terms), 0.0)
        elif isinstance(f, QuadExpr):
            off = f.aff.constant
            f = f.copy()
            f.aff.constant = 0.0
        elif isinstance(f, VariableRef):
            pass
        if spec.sense == "==":
            lb = ub = -off
        elif spec.sense == "<=":
            lb, ub = -math.inf, -off
        else:
            lb, ub = -off, math.inf
        c = ConstraintObject(f, lb, ub, restriction, name)
        self.constraints.append(c)
        self._ready = False
        return c

    def constraint_interval(self, func, lower: float, upper: float, restriction: DomainRestriction = None,
                            name: str = "") -> ConstraintObject:
        """``@constraint(m, lower <= func <= upper)`` → MOI.Interval (transform.jl:405-407)."""
        spec = ConstraintSpec(func if not is_number(func) else AffExpr(constant=func), "==")
        c = self.constraint(spec, restriction, name)
        off = -c.lb   # constant moved to the right-hand side by constraint()
        c.lb, c.ub = float(lower) - off, float(upper) - off
        return c

    def objective(self, sense: str, expr):
        assert sense in ("min", "max")
        self.objective_sense, self.objective_function = sense, expr
        self._ready = False

    # -- solve-side plumbing (delegated to the backend) ---------------------------
    def set_transformation_backend(self, backend):
        self.backend = backend
        backend._attach(self)
        self._ready = False

    # The user-facing calls of the reference's tests (test/solve.jl): each one is the InfiniteOpt / JuMP function of the
    # same name, forwarded to the transformation backend (src/infiniteopt_backend.jl).
    def transformation_backend_ready(self) -> bool:
        return bool(self._ready)

    def optimize(self):
        """``optimize!(model)``: (re)build the backend when the model changed, then ``JuMP.optimize!(backend)``."""
        if not self._ready:
            self.backend.build_transformation_backend(self)
        return self.backend.optimize()

    def set_parameter_value(self, pref, value) -> None:
        """``set_parameter_value``: θ is updated in place when the backend can (finite parameters, parameter functions:
        ``src/infiniteopt_backend.jl:511-550``); anything else leaves the backend to be rebuilt."""
        ready = self._ready
        if isinstance(pref, FiniteParameterRef):
            pref.value = float(value)
        elif isinstance(pref, ParameterFunctionRef):
            pref.func = value
        else:
            raise TypeError("set_parameter_value takes a finite parameter or a parameter function")
        self._ready = bool(ready and self.backend is not None and self.backend.update_parameter_value(pref, value))

    def set_start_value(self, vref, value) -> None:
        """``set_start_value``: written into ``core.x0`` when the backend is built (``:553-592``)."""
        ready = self._ready
        vref.info.start = value
        self._ready = bool(ready and self.backend is not None and self.backend.update_start_value(vref, value))

    def value(self, ref, label: str = "public"):
        return self.backend.map_value(ref, label)

    def dual(self, cref, label: str = "public"):
        return self.backend.map_dual(cref, label)

    def supports(self, ref, label: str = "public"):
        if isinstance(ref, ConstraintObject):
            return self.backend.constraint_supports(ref, label)
        return self.backend.variable_supports(ref, label)

    def objective_value(self) -> float:
        return self.backend.objective_value()

    def termination_status(self) -> str:
        return self.backend.termination_status()

    def primal_status(self) -> str:
        return self.backend.primal_status()

    def solve_time(self) -> float:
        return self.backend.solve_time_sec()

    def set_silent(self, value: bool = True) -> None:
        self.backend.set_silent(value)

    def set_time_limit_sec(self, value) -> None:
        self.backend.set_time_limit_sec(value)
