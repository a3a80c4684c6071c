"""Minimal JuMP-style scalar expression algebra for the modelling layer.

The reference's ``_exafy`` dispatches on the JuMP expression *type* — variable
reference, ``GenericAffExpr``, ``GenericQuadExpr``, ``GenericNonlinearExpr``
(``/root/reference/src/transform.jl:337-389``) — and walks their terms in the
insertion order of JuMP's ordered dictionaries, so the shape of the trees handed to
ExaModels (hence the COO slot order) is decided by how JuMP canonicalises
``x + y``, ``(x - d)^2``, ``u*cos(x)`` ….  This module reproduces those rules
[EXT: JuMP 1.x operators.jl / quad_expr.jl / nlp_expr.jl, restated from the public
package; not vendored under /root/reference]:

* ref/aff ± ref/aff/number → affine, terms appended in order, coefficients merged
  in place at the position of the first insertion;
* ref·ref, aff·aff → quadratic (double loop over the two term lists, then constant
  cross terms), unordered variable pairs keyed at first insertion; ``x^2`` ≡ ``x*x``;
* anything involving a nonlinear operand, a quadratic times a non-number, a
  division by a non-number or a general power → nonlinear node ``(head, args…)``.
"""
from __future__ import annotations

import numbers
from collections import OrderedDict
from typing import Callable, Iterable, List, Tuple


def is_number(x) -> bool:
    return isinstance(x, numbers.Real) and not isinstance(x, bool)


class Scalar:
    """Common arithmetic for refs and expressions."""

    __array_priority__ = 1000

    def __add__(self, o):
        return _add(self, o, 1.0)

    def __radd__(self, o):
        return _add(o, self, 1.0)

    def __sub__(self, o):
        return _add(self, o, -1.0)

    def __rsub__(self, o):
        return _add(o, self, -1.0)

    def __mul__(self, o):
        return _mul(self, o)

    def __rmul__(self, o):
        return _mul(o, self)

    def __truediv__(self, o):
        if is_number(o):
            return _mul(self, 1.0 / o)
        return NonlinearExpr("/", [self, o])

    def __rtruediv__(self, o):
        return NonlinearExpr("/", [o, self])

    def __neg__(self):
        if isinstance(self, NonlinearExpr):
            return NonlinearExpr("-", [self])
        return _mul(-1.0, self)

    def __pos__(self):
        return self

    def __pow__(self, p):
        if is_number(p) and not isinstance(self, NonlinearExpr):
            if p == 2 and not isinstance(self, QuadExpr):
                return _mul(self, self)
            if p == 1:
                return self
            if p == 0:
                return 1.0
        return NonlinearExpr("^", [self, p])

    def __rpow__(self, b):
        return NonlinearExpr("^", [b, self])

    # comparison operators build (function, set) pairs for constraints
    def __eq__(self, o):  # type: ignore[override]
        return ConstraintSpec(_add(self, o, -1.0), "==")

    def __le__(self, o):
        return ConstraintSpec(_add(self, o, -1.0), "<=")

    def __ge__(self, o):
        return ConstraintSpec(_add(self, o, -1.0), ">=")

    __hash__ = object.__hash__


class VariableRef(Scalar):
    """``InfiniteOpt.GeneralVariableRef`` analogue; identity-hashed."""

    def __init__(self, model, kind: str, name: str = ""):
        self.model, self.kind, self.name = model, kind, name

    def __eq__(self, o):  # type: ignore[override]
        if o is None:
            return False
        if self is o:
            return True
        spec = Scalar.__eq__(self, o)
        # ref == ref doubles as the (identity) equality test of containers: falsy there
        spec._truth = False if isinstance(o, VariableRef) else None
        return spec

    __hash__ = object.__hash__

    def __repr__(self):
        return self.name or f"<{self.kind}>"


class AffExpr(Scalar):
    def __init__(self, terms: "OrderedDict[VariableRef, float]" = None, constant: float = 0.0):
        self.terms: "OrderedDict[VariableRef, float]" = terms if terms is not None else OrderedDict()
        self.constant = float(constant)

    def copy(self) -> "AffExpr":
        return AffExpr(OrderedDict(self.terms), self.constant)

    def add_term(self, coef: float, v: VariableRef):
        if coef == 0.0:   # JuMP._add_or_set!: adding a zero term is a no-op
            return
        self.terms[v] = self.terms.get(v, 0.0) + coef

    def linear_terms(self) -> Iterable[Tuple[float, VariableRef]]:
        return [(c, v) for v, c in self.terms.items()]

    def is_zero(self) -> bool:
        return self.constant == 0.0 and all(c == 0.0 for c in self.terms.values())

    def __repr__(self):
        s = " + ".join(f"{c}*{v!r}" for v, c in self.terms.items())
        return f"({s} + {self.constant})"


class _Pair:
    """``JuMP.UnorderedPair``."""

    __slots__ = ("a", "b")

    def __init__(self, a, b):
        self.a, self.b = a, b

    def __hash__(self):
        return hash(frozenset((id(self.a), id(self.b))))

    def __eq__(self, o):
        return (self.a is o.a and self.b is o.b) or (self.a is o.b and self.b is o.a)


class QuadExpr(Scalar):
    def __init__(self, aff: AffExpr = None, terms: "OrderedDict[_Pair, float]" = None):
        self.aff = aff if aff is not None else AffExpr()
        self.terms: "OrderedDict[_Pair, float]" = terms if terms is not None else OrderedDict()

    def copy(self) -> "QuadExpr":
        return QuadExpr(self.aff.copy(), OrderedDict(self.terms))

    def add_quad(self, coef: float, a: VariableRef, b: VariableRef):
        if coef == 0.0:
            return
        k = _Pair(a, b)
        self.terms[k] = self.terms.get(k, 0.0) + coef

    def quad_terms(self) -> List[Tuple[float, VariableRef, VariableRef]]:
        return [(c, k.a, k.b) for k, c in self.terms.items()]

    def __repr__(self):
        s = " + ".join(f"{c}*{k.a!r}*{k.b!r}" for k, c in self.terms.items())
        return f"[{s} + {self.aff!r}]"


class NonlinearExpr(Scalar):
    def __init__(self, head: str, args: list):
        self.head, self.args = head, list(args)

    def __repr__(self):
        return f"{self.head}({', '.join(repr(a) for a in self.args)})"


class ConstraintSpec:
    """``expr (==|<=|>=) 0`` before normalisation into (function, MOI set)."""

    def __init__(self, func, sense: str):
        self.func, self.sense = func, sense
        self._truth = None

    def __bool__(self):
        if self._truth is not None:
            return self._truth
        raise TypeError("a constraint is not a boolean; pass it to model.constraint(...)")


# --------------------------------------------------------------------------
def _to_aff(x) -> AffExpr:
    if isinstance(x, AffExpr):
        return x.copy()
    if isinstance(x, VariableRef):
        return AffExpr(OrderedDict([(x, 1.0)]))
    if is_number(x):
        return AffExpr(constant=x)
    raise TypeError(type(x))


def _add(a, b, sign: float):
    """a + sign*b."""
    if isinstance(a, NonlinearExpr) or isinstance(b, NonlinearExpr):
        if is_number(b) and b == 0:
            return a
        if is_number(a) and a == 0 and sign > 0:
            return b
        return NonlinearExpr("+" if sign > 0 else "-", [a, b])
    if isinstance(a, QuadExpr) or isinstance(b, QuadExpr):
        out = a.copy() if isinstance(a, QuadExpr) else QuadExpr(_to_aff(a))
        if isinstance(b, QuadExpr):
            for c, v1, v2 in b.quad_terms():
                out.add_quad(sign * c, v1, v2)
            for c, v in b.aff.linear_terms():
                out.aff.add_term(sign * c, v)
            out.aff.constant += sign * b.aff.constant
        else:
            bb = _to_aff(b)
            for c, v in bb.linear_terms():
                out.aff.add_term(sign * c, v)
            out.aff.constant += sign * bb.constant
        return out
    out = _to_aff(a)
    bb = _to_aff(b)
    for c, v in bb.linear_terms():
        out.add_term(sign * c, v)
    out.constant += sign * bb.constant
    return out


def _mul(a, b):
    if is_number(a) and is_number(b):
        return a * b
    if isinstance(a, NonlinearExpr) or isinstance(b, NonlinearExpr):
        return NonlinearExpr("*", [a, b])
    if is_number(a) or is_number(b):
        c, e = (a, b) if is_number(a) else (b, a)
        c = float(c)
        if isinstance(e, VariableRef):
            return AffExpr(OrderedDict([(e, c)]))
        if isinstance(e, AffExpr):
            return AffExpr(OrderedDict((v, c * k) for v, k in e.terms.items()), c * e.constant)
        out = QuadExpr(_mul(c, e.aff), OrderedDict((k, c * v) for k, v in e.terms.items()))
        return out
    if isinstance(a, QuadExpr) or isinstance(b, QuadExpr):
        return NonlinearExpr("*", [a, b])
    la, lb = _to_aff(a), _to_aff(b)
    out = QuadExpr()
    for ca, va in la.linear_terms():
        for cb, vb in lb.linear_terms():
            out.add_quad(ca * cb, va, vb)
    if la.constant != 0.0:
        for cb, vb in lb.linear_terms():
            out.aff.add_term(la.constant * cb, vb)
    if lb.constant != 0.0:
        for ca, va in la.linear_terms():
            out.aff.add_term(lb.constant * ca, va)
    out.aff.constant = la.constant * lb.constant
    return out


def nl(head: str) -> Callable:
    """Nonlinear univariate operator usable on refs/expressions (``sin(x)`` …)."""

    def fn(x):
        if is_number(x):
            from .nodes import _SCALAR
            return _SCALAR[head](x)
        return NonlinearExpr(head, [x])

    fn.__name__ = head
    return fn


def all_expression_variables(expr) -> List[VariableRef]:
    """``InfiniteOpt.all_expression_variables``: unique refs in first-seen order."""
    seen, out = set(), []

    def visit(e):
        if isinstance(e, VariableRef):
            if id(e) not in seen:
                seen.add(id(e))
                out.append(e)
        elif isinstance(e, AffExpr):
            for v in e.terms:
                visit(v)
        elif isinstance(e, QuadExpr):
            for k in e.terms:
                visit(k.a)
                visit(k.b)
            visit(e.aff)
        elif isinstance(e, NonlinearExpr):
            for a in e.args:
                visit(a)

    visit(expr)
    return out


def map_expression(transform: Callable, expr):
    """``InfiniteOpt.map_expression``: rebuild ``expr`` with every ref replaced."""
    if isinstance(expr, VariableRef):
        return transform(expr)
    if isinstance(expr, AffExpr):
        out = 0.0
        first = True
        for c, v in expr.linear_terms():
            term = _mul(c, transform(v))
            out = term if first else _add(out, term, 1.0)
            first = False
        return _add(out, expr.constant, 1.0) if not first else expr.constant
    if isinstance(expr, QuadExpr):
        out = None
        for c, a, b in expr.quad_terms():
            term = _mul(c, _mul(transform(a), transform(b)))
            out = term if out is None else _add(out, term, 1.0)
        rest = map_expression(transform, expr.aff)
        return rest if out is None else _add(out, rest, 1.0)
    if isinstance(expr, NonlinearExpr):
        return NonlinearExpr(expr.head, [map_expression(transform, a) for a in expr.args])
    return expr
