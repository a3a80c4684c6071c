"""Operator table: JuMP head symbol → template-node constructor.

Mirror of ``/root/reference/src/operators.jl:2-54``.  The table is reproduced entry
for entry, including the reference's ``:csch => csc`` mapping (``operators.jl:40``),
which sends the *hyperbolic* cosecant head to the *trigonometric* cosecant — a latent
reference quirk that a drop-in must not silently change.
Operators absent from the reference table (``asinh``, ``acosh``, ``min`` …,
``operators.jl:45`` TODO) raise the same error text.
"""
from __future__ import annotations

from . import nodes as N


def _fold(op):
    return lambda *args: N.nary(op, *args)


_op_mappings = {
    "+": _fold("+"),
    "-": _fold("-"),
    "*": _fold("*"),
    "/": _fold("/"),
    "^": _fold("^"),
}
for _name in ("inv", "sqrt", "cbrt", "abs", "abs2", "exp", "exp2", "log", "log2", "log10", "log1p",
              "sin", "cos", "tan", "asin", "acos", "csc", "sec", "cot", "atan", "acot",
              "sind", "cosd", "tand", "cscd", "secd", "cotd", "atand", "acotd",
              "sinh", "cosh", "tanh", "sech", "coth", "atanh", "acoth"):
    _op_mappings[_name] = N.FUNCS[_name]
_op_mappings["csch"] = N.FUNCS["csc"]   # operators.jl:40 — reproduced as is


def nl_op(s: str):
    """``_nl_op`` (operators.jl:49-54)."""
    if s not in _op_mappings:
        raise KeyError(f"`InfiniteExaModel`s does not support the nonlinear operator `{s}`. "
                       "If you need support for this operator, please open an issue.")
    return _op_mappings[s]
