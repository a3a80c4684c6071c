"""Expression-tree vocabulary of a SIMD template.

Host-side mirror of the node types the reference's transcriber emits into
ExaModels (``/root/reference/src/transform.jl:290-389``): ``Var[idx]``,
``ParameterNode``, item-data leaves taken from a ``DataSource``, real constants
captured in binary nodes, the unary operators of ``src/operators.jl:3-44`` and the
binary ``+ - * / ^``.  N-ary ``+``/``*`` are folded left exactly as Julia's
``afoldl`` does for ``transform.jl:388``.

Only tree *construction* lives here; evaluation happens on the device (HIP) and,
for tests, in ``oracle/``.
"""
from __future__ import annotations

import math
import numbers
from typing import Dict, Tuple, Union

Number = Union[int, float]

# opcode table — must match include/iem_blob.h
OP = {
    "const": 0, "data": 1, "par": 2, "var": 3,
    "+": 10, "-": 11, "*": 12, "/": 13, "^": 14,
    "neg": 20, "pos": 21, "inv": 22, "sqrt": 23, "cbrt": 24, "abs": 25, "abs2": 26,
    "exp": 27, "exp2": 28, "log": 29, "log2": 30, "log10": 31, "log1p": 32,
    "sin": 33, "cos": 34, "tan": 35, "asin": 36, "acos": 37, "csc": 38, "sec": 39,
    "cot": 40, "atan": 41, "acot": 42, "sind": 43, "cosd": 44, "tand": 45, "cscd": 46,
    "secd": 47, "cotd": 48, "atand": 49, "acotd": 50, "sinh": 51, "cosh": 52,
    "tanh": 53, "csch": 54, "sech": 55, "coth": 56, "atanh": 57, "acoth": 58,
}
OP_NAME = {v: k for k, v in OP.items()}
BINARY_OPS = ("+", "-", "*", "/", "^")
UNARY_OPS = tuple(k for k, v in OP.items() if 20 <= v < 59)


def _is_real(x) -> bool:
    return isinstance(x, numbers.Real) and not isinstance(x, bool)


class Node:
    """Base class of every tree node (ExaModels.AbstractNode analogue)."""

    __slots__ = ()
    __array_priority__ = 1000  # keep numpy scalars from hijacking arithmetic

    # binary arithmetic -------------------------------------------------
    def __add__(self, o):
        return _binary("+", self, o)

    def __radd__(self, o):
        return _binary("+", o, self)

    def __sub__(self, o):
        return _binary("-", self, o)

    def __rsub__(self, o):
        return _binary("-", o, self)

    def __mul__(self, o):
        return _binary("*", self, o)

    def __rmul__(self, o):
        return _binary("*", o, self)

    def __truediv__(self, o):
        return _binary("/", self, o)

    def __rtruediv__(self, o):
        return _binary("/", o, self)

    def __pow__(self, o):
        return _binary("^", self, o)

    def __rpow__(self, o):
        return _binary("^", o, self)

    def __neg__(self):
        return Unary("neg", self)

    def __pos__(self):
        return Unary("pos", self)

    def __eq__(self, o):  # structural equality (Julia's egal on immutable nodes)
        return type(self) is type(o) and self._key() == o._key()

    def __hash__(self):
        return hash((type(self).__name__, self._key()))

    def _key(self):
        raise NotImplementedError


class Const(Node):
    __slots__ = ("value", "_int")

    def __init__(self, value: Number):
        self.value = float(value)
        self._int = isinstance(value, numbers.Integral)

    def _key(self):
        return (self.value,)

    def __repr__(self):
        return repr(self.value)


class DataField(Node):
    """``data_src[:name]`` — one field of the item NamedTuple (ExaModels ParIndexed).

    Usable both as a Float64 leaf (``transform.jl:320-322``) and inside a variable
    index (``transform.jl:308-311, 471-506``)."""

    __slots__ = ("name",)

    def __init__(self, name: str):
        self.name = name

    def _key(self):
        return (self.name,)

    def __repr__(self):
        return f"i.{self.name}"


class DataSource:
    """``ExaModels.DataSource()`` analogue: indexing yields item-field leaves."""

    def __getitem__(self, name: str) -> DataField:
        return DataField(str(name))

    def __getattr__(self, name: str) -> DataField:
        if name.startswith("__"):
            raise AttributeError(name)
        return DataField(name)


class Var(Node):
    """``x[i]`` with ``i`` a 1-based literal or an index expression over item fields."""

    __slots__ = ("i",)

    def __init__(self, i):
        self.i = i

    def _key(self):
        return (_idx_key(self.i),)

    def __repr__(self):
        return f"x[{self.i!r}]"


class ParameterNode(Node):
    """``θ[i]`` — entry of the parameter vector (ExaModels.ParameterNode)."""

    __slots__ = ("i",)

    def __init__(self, i):
        self.i = i

    def _key(self):
        return (_idx_key(self.i),)

    def __repr__(self):
        return f"θ[{self.i!r}]"


class Unary(Node):
    __slots__ = ("op", "inner")

    def __init__(self, op: str, inner: Node):
        assert op in UNARY_OPS, op
        self.op = op
        self.inner = inner

    def _key(self):
        return (self.op, self.inner)

    def __repr__(self):
        return f"{self.op}({self.inner!r})"


class Binary(Node):
    __slots__ = ("op", "inner1", "inner2")

    def __init__(self, op: str, inner1: Node, inner2: Node):
        assert op in BINARY_OPS, op
        self.op = op
        self.inner1 = inner1
        self.inner2 = inner2

    def _key(self):
        return (self.op, self.inner1, self.inner2)

    def __repr__(self):
        return f"({self.inner1!r} {self.op} {self.inner2!r})"


class Null:
    """``ExaModels.Null(c)`` — a constant template (``transform.jl:392-393``)."""

    __slots__ = ("value",)

    def __init__(self, value: Number = 0.0):
        self.value = float(value)

    def __repr__(self):
        return f"Null({self.value})"


def _wrap(x) -> Node:
    if isinstance(x, Node):
        return x
    if _is_real(x):
        return Const(x)
    raise TypeError(f"cannot use {type(x).__name__} in a template expression")


def _binary(op: str, a, b) -> Node:
    return Binary(op, _wrap(a), _wrap(b))


def _idx_key(i):
    if isinstance(i, Node):
        c0, terms = affine_index(i)
        return (c0, tuple(sorted(terms.items())))
    return (int(i), ())


def affine_index(node) -> Tuple[int, Dict[str, int]]:
    """Reduce an index expression to ``c0 + Σ coef·field`` over integer item fields.

    The reference builds such expressions at ``transform.jl:471-506`` (``idx`` is
    ``data_src[:group_idxK]`` ± an integer) and ExaModels' ``Variable`` indexing adds
    the slab offset and the column-major strides."""
    if isinstance(node, numbers.Integral):
        return int(node), {}
    if isinstance(node, Const):
        if node.value != int(node.value):
            raise ValueError("non-integer constant in an index expression")
        return int(node.value), {}
    if isinstance(node, DataField):
        return 0, {node.name: 1}
    if isinstance(node, Unary) and node.op in ("neg", "pos"):
        c0, t = affine_index(node.inner)
        s = -1 if node.op == "neg" else 1
        return s * c0, {k: s * v for k, v in t.items()}
    if isinstance(node, Binary) and node.op in ("+", "-"):
        c1, t1 = affine_index(node.inner1)
        c2, t2 = affine_index(node.inner2)
        s = 1 if node.op == "+" else -1
        out = dict(t1)
        for k, v in t2.items():
            out[k] = out.get(k, 0) + s * v
        return c1 + s * c2, {k: v for k, v in out.items() if v != 0}
    if isinstance(node, Binary) and node.op == "*":
        c1, t1 = affine_index(node.inner1)
        c2, t2 = affine_index(node.inner2)
        if t1 and t2:
            raise ValueError("index expression is not affine in the item fields")
        if not t1:
            c1, t1, c2, t2 = c2, t2, c1, t1
        return c1 * c2, {k: v * c2 for k, v in t1.items() if v * c2 != 0}
    raise ValueError(f"unsupported index expression {node!r}")


# ---------------------------------------------------------------------------
# operator functions (the right-hand sides of src/operators.jl:3-44)
# ---------------------------------------------------------------------------

def _unary_fn(name: str, scalar):
    def fn(x):
        if isinstance(x, Node):
            return Unary(name, x)
        return scalar(x)

    fn.__name__ = name
    fn.__doc__ = f"`{name}` on a template node (or a plain float)."
    return fn


def _deg(f):
    return lambda x: f(math.radians(x))


_SCALAR = {
    "inv": lambda x: 1.0 / x,
    "sqrt": math.sqrt,
    "cbrt": lambda x: math.copysign(abs(x) ** (1.0 / 3.0), x),
    "abs": abs,
    "abs2": lambda x: x * x,
    "exp": math.exp,
    "exp2": lambda x: 2.0 ** x,
    "log": math.log,
    "log2": math.log2,
    "log10": math.log10,
    "log1p": math.log1p,
    "sin": math.sin,
    "cos": math.cos,
    "tan": math.tan,
    "asin": math.asin,
    "acos": math.acos,
    "csc": lambda x: 1.0 / math.sin(x),
    "sec": lambda x: 1.0 / math.cos(x),
    "cot": lambda x: 1.0 / math.tan(x),
    "atan": math.atan,
    "acot": lambda x: math.atan(1.0 / x),
    "sind": _deg(math.sin),
    "cosd": _deg(math.cos),
    "tand": _deg(math.tan),
    "cscd": lambda x: 1.0 / math.sin(math.radians(x)),
    "secd": lambda x: 1.0 / math.cos(math.radians(x)),
    "cotd": lambda x: 1.0 / math.tan(math.radians(x)),
    "atand": lambda x: math.degrees(math.atan(x)),
    "acotd": lambda x: math.degrees(math.atan(1.0 / x)),
    "sinh": math.sinh,
    "cosh": math.cosh,
    "tanh": math.tanh,
    "csch": lambda x: 1.0 / math.sinh(x),
    "sech": lambda x: 1.0 / math.cosh(x),
    "coth": lambda x: 1.0 / math.tanh(x),
    "atanh": math.atanh,
    "acoth": lambda x: math.atanh(1.0 / x),
}

FUNCS = {name: _unary_fn(name, f) for name, f in _SCALAR.items()}
globals().update(FUNCS)


def nary(op: str, *args):
    """Left fold of an n-ary ``+``/``*`` (Julia ``afoldl``; ``transform.jl:388``)."""
    if op in FUNCS or op in ("neg", "pos"):
        (a,) = args
        if op == "neg":
            return -a
        if op == "pos":
            return +a
        return FUNCS[op](a)
    if op == "-" and len(args) == 1:
        return -args[0]
    if op == "+" and len(args) == 1:
        return +args[0]
    acc = args[0]
    for a in args[1:]:
        if op == "+":
            acc = acc + a
        elif op == "-":
            acc = acc - a
        elif op == "*":
            acc = acc * a
        elif op == "/":
            acc = acc / a
        elif op == "^":
            acc = acc ** a
        else:
            raise KeyError(op)
    return acc
