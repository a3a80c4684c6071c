"""``ExaCore`` builder: the four calls the reference's transcriber makes.

Mirror of the ExaModels surface used by ``/root/reference/src/transform.jl``:
``add_var`` (``:113,154``), ``add_par`` (``:127,179``), ``add_con``
(``:458,559,597``), ``add_obj`` (``:614,700,741``), the ``Variable``/``Parameter``
handles with ``.offset/.length/.size`` (``src/infiniteopt_backend.jl:476-479,560``)
and ``Var.i`` (``:562``), and ``set_parameter!`` (``:522,546``).

The core only *records* templates; :func:`ExaCore.to_blob` serialises them into the
wire format of ``include/iem_blob.h`` which ``libiem_hip.so`` (and, in tests, the
oracle) consume.  Offsets follow ExaModels' call-order bookkeeping: a template's
row offset ``o0`` and its COO offsets ``o1``/``o2`` are the running counters at the
moment ``add_con``/``add_obj`` is called.
"""
from __future__ import annotations

import hashlib
import numbers
import struct
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from .items import Items, as_items, Field
from .nodes import (OP, Binary, Const, DataField, Node, Null, ParameterNode, Unary, Var,
                    affine_index)

BLOB_MAGIC = 0x31424F4C424D4549
BLOB_VERSION = 1
MAX_IDX_TERMS = 3
A_F64_DATA, A_I64_DATA, A_F64_FILL, A_I64_RANGE = 0, 1, 2, 3
T_OBJ, T_CON = 0, 1
F_AFFINE, F_GATHER = 0, 1


def _f2w(x: float) -> int:
    return struct.unpack("<q", struct.pack("<d", float(x)))[0]


class Variable:
    """``ExaModels.Variable``: a contiguous slab, first index fastest
    (``/root/reference/test/transcription.jl:44-57,164``)."""

    def __init__(self, size: Tuple[int, ...], offset: int):
        self.size = tuple(int(n) for n in size)
        self.length = int(np.prod(self.size)) if self.size else 1
        self.offset = int(offset)

    def _lin(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        if len(idx) != len(self.size):
            raise IndexError(f"expected {len(self.size)} indices, got {len(idx)}")
        lin, stride = self.offset, 1   # 1-based: offset + i1 + Σ_{d>1} (i_d - 1)·stride_d
        for d, (i, n) in enumerate(zip(idx, self.size)):
            if isinstance(i, numbers.Integral) and not 1 <= i <= n:
                raise IndexError(f"index {i} out of bounds 1:{n}")
            if isinstance(i, numbers.Integral):
                i = int(i)
            lin = lin + (i if d == 0 else (i - 1) * stride)
            stride *= n
        return lin

    def __getitem__(self, idx) -> Var:
        return Var(self._lin(idx))


class Parameter(Variable):
    """``ExaModels.Parameter``: slab of θ (``add_par``)."""

    def __getitem__(self, idx) -> ParameterNode:
        return ParameterNode(self._lin(idx))


class Constraint:
    def __init__(self, tpl_index: int, offset: int, length: int):
        self.tpl_index, self.offset, self.length = tpl_index, offset, length


class Objective:
    def __init__(self, tpl_index: int):
        self.tpl_index = tpl_index


class _Template:
    __slots__ = ("kind", "items", "nodes", "root", "ifields", "ffields", "idx", "lcon", "ucon",
                 "o0", "expr", "tag")


class _Grow:
    """Append-only float64 buffer with capacity doubling; ``view`` is the live (writable) prefix."""

    def __init__(self):
        self.buf = np.zeros(0)
        self.n = 0

    def append(self, a: np.ndarray) -> None:
        need = self.n + a.size
        if need > self.buf.size:
            nb = np.empty(max(need, 2 * self.buf.size, 1024))
            nb[:self.n] = self.buf[:self.n]
            self.buf = nb
        self.buf[self.n:need] = a
        self.n = need

    def append_fill(self, n: int, value: float) -> None:
        need = self.n + n
        if need > self.buf.size:
            nb = np.empty(max(need, 2 * self.buf.size, 1024))
            nb[:self.n] = self.buf[:self.n]
            self.buf = nb
        self.buf[self.n:need] = value        # no temporary: fresh pages are touched once
        self.n = need

    def reserve(self, extra: int) -> None:
        if self.n + extra > self.buf.size:
            nb = np.empty(self.n + extra)
            nb[:self.n] = self.buf[:self.n]
            self.buf = nb

    @property
    def view(self) -> np.ndarray:
        return self.buf[:self.n]


class ExaCore:
    """``ExaModels.ExaCore(; backend, minimize, concrete)`` (``transform.jl:815``)."""

    def __init__(self, backend=None, minimize: bool = True, concrete=True):
        self.backend = backend
        self.minimize = bool(minimize)
        # x0 / lvar / uvar / theta grow slab by slab: amortised-doubling buffers (np.concatenate per
        # add_var copies everything built so far — 5 s of the 10^6-support quadrotor's 8.6 s build)
        self._bufs = {k: _Grow() for k in ("x0", "lvar", "uvar", "theta")}
        self.templates: List[_Template] = []
        self.ncon = 0
        self._model = None  # live device model, for set_parameter!
        # (offset, dims, infinite-parameter group of each axis; 0 = none) of every add_var slab — the
        # blob's slab table, from which iem_create_sharded cuts a rank's window (include/iem_blob.h)
        self.slabs: List[Tuple[int, Tuple[int, ...], Tuple[int, ...]]] = []

    # the core buffers (views into the growable storage; element writes go through)
    x0 = property(lambda self: self._bufs["x0"].view)
    lvar = property(lambda self: self._bufs["lvar"].view)
    uvar = property(lambda self: self._bufs["uvar"].view)
    theta = property(lambda self: self._bufs["theta"].view)

    # -- sizes -----------------------------------------------------------
    @property
    def nvar(self) -> int:
        return self.x0.shape[0]

    @property
    def npar(self) -> int:
        return self.theta.shape[0]

    # -- builders ------------------------------------------------------------
    def reserve_vars(self, n: int) -> None:
        """Capacity hint: `n` more variables are about to be added (no semantic effect)."""
        for k in ("x0", "lvar", "uvar"):
            self._bufs[k].reserve(int(n))

    def add_var(self, *dims: int, start=0.0, lvar=-np.inf, uvar=np.inf, groups=None) -> Variable:
        """``groups``: the infinite-parameter group (1-based, as in the items' grid hint) each axis of the
        slab runs over — bookkeeping for sharding only, no effect on evaluation."""
        dims = tuple(int(d) for d in dims) or (1,)
        n = int(np.prod(dims))
        var = Variable(dims, self.nvar)
        groups = tuple(int(g) for g in groups) if groups is not None else (0,) * len(dims)
        assert len(groups) == len(dims)
        if len(dims) <= 3:
            self.slabs.append((var.offset, dims, groups))

        def expand(v):
            a = np.asarray(v, dtype=np.float64)
            if a.ndim == 0:
                return np.full(n, float(a))
            # Julia arrays are column-major: element [i1,i2] sits at i1 + n1*(i2-1)
            assert a.shape == dims, (a.shape, dims)
            return np.ascontiguousarray(a.reshape(-1, order="F"))

        for key, v in (("x0", start), ("lvar", lvar), ("uvar", uvar)):
            if np.ndim(v) == 0:
                self._bufs[key].append_fill(n, float(v))
            else:
                self._bufs[key].append(expand(v))
        return var

    def add_par(self, vals) -> Parameter:
        a = np.asarray(vals, dtype=np.float64)
        dims = a.shape if a.ndim else (1,)
        par = Parameter(dims, self.npar)
        self._bufs["theta"].append(a.reshape(-1, order="F"))
        return par

    def add_con(self, expr, itr=None, lcon=0.0, ucon=0.0) -> Constraint:
        items = as_items(itr if itr is not None else [dict()])
        t = self._compile(T_CON, expr, items)
        n = len(items)
        t.lcon = self._bound(lcon, n)
        t.ucon = self._bound(ucon, n)
        t.o0 = self.ncon
        self.ncon += n
        self.templates.append(t)
        return Constraint(len(self.templates) - 1, t.o0, n)

    def add_obj(self, expr, itr=None) -> Objective:
        items = as_items(itr if itr is not None else [dict()])
        t = self._compile(T_OBJ, expr, items)
        t.lcon = t.ucon = 0.0
        t.o0 = 0
        self.templates.append(t)
        return Objective(len(self.templates) - 1)

    def set_parameter(self, par: Parameter, vals) -> None:
        """``ExaModels.set_parameter!(core, param, vals)`` (``infiniteopt_backend.jl:522,546``)."""
        a = np.asarray(vals, dtype=np.float64).reshape(-1, order="F")
        if a.shape[0] != par.length:
            raise ValueError("parameter length mismatch")
        self.theta[par.offset:par.offset + par.length] = a
        if self._model is not None:
            self._model.set_parameter(par.offset, a)

    # -- helpers ---------------------------------------------------------
    @staticmethod
    def _bound(v, n):
        a = np.asarray(v, dtype=np.float64)
        if a.ndim == 0:
            return float(a)
        a = a.reshape(-1, order="F")
        assert a.shape[0] == n
        return np.ascontiguousarray(a)

    def _compile(self, kind: int, expr, items: Items) -> _Template:
        if isinstance(expr, numbers.Real):
            expr = Null(expr)
        t = _Template()
        t.kind, t.items, t.expr = kind, items, expr
        t.tag = ("tpl", len(self.templates))   # the transcriber overwrites it with a model-level tag
        t.nodes, t.ifields, t.ffields, t.idx = [], [], [], []
        if_ids: Dict[str, int] = {}
        ff_ids: Dict[str, int] = {}
        idx_ids: Dict[tuple, int] = {}

        def ifield(name: str) -> int:
            if name not in if_ids:
                f = items.fields.get(name)
                if f is None:
                    raise KeyError(f"item iterator has no field `{name}`")
                if f.kind != "int":
                    raise TypeError(f"item field `{name}` is not an integer; cannot index with it")
                if_ids[name] = len(t.ifields)
                t.ifields.append(f)
            return if_ids[name]

        def ffield(name: str) -> int:
            if name not in ff_ids:
                f = items.fields.get(name)
                if f is None:
                    raise KeyError(f"item iterator has no field `{name}`")
                if f.kind == "int":  # integer item data used as a Float64 leaf
                    vals = f.values(items.dims).astype(np.float64)
                    steps, s = [], 1
                    for n in items.dims:
                        steps.append(s)
                        s *= n
                    f = Field("float", "gather", 0, tuple(steps), np.ascontiguousarray(vals))
                ff_ids[name] = len(t.ffields)
                t.ffields.append(f)
            return ff_ids[name]

        def index(i) -> int:
            c0, terms = affine_index(i)
            if len(terms) > MAX_IDX_TERMS:
                raise ValueError("index expression uses too many item fields")
            key = (c0, tuple((ifield(k), v) for k, v in terms.items()))
            if key not in idx_ids:
                idx_ids[key] = len(t.idx)
                t.idx.append(key)
            return idx_ids[key]

        def emit(node) -> int:
            if isinstance(node, Const):
                t.nodes.append((OP["const"], 0, 0, node.value))
            elif isinstance(node, DataField):
                t.nodes.append((OP["data"], ffield(node.name), 0, 0.0))
            elif isinstance(node, ParameterNode):
                t.nodes.append((OP["par"], index(node.i), 0, 0.0))
            elif isinstance(node, Var):
                t.nodes.append((OP["var"], index(node.i), 0, 0.0))
            elif isinstance(node, Unary):
                a = emit(node.inner)
                t.nodes.append((OP[node.op], a, 0, 0.0))
            elif isinstance(node, Binary):
                a = emit(node.inner1)
                b = emit(node.inner2)
                t.nodes.append((OP[node.op], a, b, 0.0))
            else:
                raise TypeError(f"not a template expression: {node!r}")
            return len(t.nodes) - 1

        if isinstance(expr, Null):
            t.nodes.append((OP["const"], 0, 0, expr.value))
            t.root = 0
        else:
            t.root = emit(expr)
        return t

    # -- serialisation -------------------------------------------------------
    def bounds(self) -> Tuple[np.ndarray, np.ndarray]:
        """Assembled ``lcon``/``ucon`` (host mirrors; ``meta.lcon``/``meta.ucon``)."""
        lcon = np.zeros(self.ncon)
        ucon = np.zeros(self.ncon)
        for t in self.templates:
            if t.kind != T_CON:
                continue
            n = len(t.items)
            lcon[t.o0:t.o0 + n] = t.lcon
            ucon[t.o0:t.o0 + n] = t.ucon
        return lcon, ucon

    def to_blob(self) -> bytes:
        arrays: List[tuple] = []   # (kind, n, payload ndarray | None, a, b)
        dedupe: Dict[tuple, int] = {}

        def add_array(a: np.ndarray) -> int:
            a = np.ascontiguousarray(a)
            if a.dtype == np.float64 and a.size and np.all(a == a.flat[0]) and not np.isnan(a.flat[0]):
                key = ("fill", a.size, _f2w(a.flat[0]))
                if key not in dedupe:
                    dedupe[key] = len(arrays)
                    arrays.append((A_F64_FILL, a.size, None, _f2w(a.flat[0]), 0))
                return dedupe[key]
            kind = A_F64_DATA if a.dtype == np.float64 else A_I64_DATA
            assert a.dtype in (np.float64, np.int64)
            h = hashlib.blake2b(a.tobytes(), digest_size=16).digest()
            hkey = ("data", kind, a.size, h)
            if hkey not in dedupe:
                dedupe[hkey] = len(arrays)
                arrays.append((kind, a.size, a, 0, 0))
            return dedupe[hkey]

        core_ids = [add_array(self.x0), add_array(self.lvar), add_array(self.uvar), add_array(self.theta)]

        tpl_words: List[List[int]] = []
        for t in self.templates:
            w: List[int] = []
            dims = list(t.items.dims) + [1] * (3 - len(t.items.dims))
            nd = len(t.items.dims)
            grid = t.items.grid
            if grid is not None and len(grid[0]) == nd:
                gid = 0
                for g in grid[0]:
                    gid = gid * 4096 + (int(g) + 1)   # group ids (and virtual collocation grids) < 4095
                origin = list(grid[1]) + [0] * (3 - nd)
            elif grid is not None and len(grid[0]) == 0:
                gid, origin = 0, [0, 0, 0]
            else:
                gid, origin = -1, [0, 0, 0]
            w += [t.kind, len(t.items), nd] + dims + [gid] + origin
            w += [len(t.ifields), len(t.ffields), len(t.idx), len(t.nodes), t.root]
            for b in (t.lcon, t.ucon):
                if isinstance(b, np.ndarray):
                    w += [1, 0, add_array(b)]
                else:
                    w += [0, _f2w(b), -1]
            for f in t.ifields + t.ffields:
                steps = list(f.steps) + [0] * (3 - len(f.steps))
                if f.mode == "affine":
                    w += [F_AFFINE, f.base] + steps + [-1]
                else:
                    w += [F_GATHER, f.base] + steps + [add_array(f.arr)]
            for c0, terms in t.idx:
                w += [c0, len(terms)]
                for fid, coef in terms:
                    w += [fid, coef]
                w += [0, 0] * (MAX_IDX_TERMS - len(terms))
            for op, a, b, imm in t.nodes:
                w += [op, a, b, _f2w(imm)]
            tpl_words.append(w)

        n_arr, n_tpl = len(arrays), len(self.templates)
        pos = 14 + 6 * n_arr + n_tpl
        tpl_off = []
        for w in tpl_words:
            tpl_off.append(pos)
            pos += len(w)
        # slab table (optional section, header word 9): present when the recorded slabs tile 0..nvar
        slab_words: List[int] = []
        cover = 0
        for off, dims, groups in self.slabs:
            if off != cover:
                break
            cover += int(np.prod(dims))
        if self.slabs and cover == self.nvar:
            slab_words = [len(self.slabs)]
            for off, dims, groups in self.slabs:
                slab_words += [off, len(dims)] + list(dims) + [1] * (3 - len(dims)) + list(groups) + [0] * (3 - len(dims))
        slab_off = pos if slab_words else 0
        pos += len(slab_words)
        arr_off = []
        for kind, n, payload, a, b in arrays:
            arr_off.append(pos if payload is not None else 0)
            if payload is not None:
                pos += n
        total = pos

        head = [BLOB_MAGIC, BLOB_VERSION, self.nvar, self.npar, self.ncon, n_tpl, n_arr,
                1 if self.minimize else 0, total, slab_off] + core_ids
        for (kind, n, payload, a, b), off in zip(arrays, arr_off):
            head += [kind, n, off, a, b, 0]
        head += tpl_off
        for w in tpl_words:
            head += w
        head += slab_words
        parts = [np.asarray(head, dtype=np.int64).tobytes()]
        for kind, n, payload, a, b in arrays:
            if payload is not None:
                parts.append(payload.tobytes())
        blob = b"".join(parts)
        assert len(blob) == 8 * total
        self._blob_arrays = arrays   # array table of the last blob (tooling/tests)
        return blob
