"""Item iterators of a SIMD template.

The reference hands ExaModels a ``Vector{NamedTuple}`` per template
(``/root/reference/src/transform.jl:31, 440-451, 538-544, 584-591, 632, 670``): one
record per item with integer support indices (``group_idxK``, ``i1``/``i2``),
Float64 support values (``ip…``/``dp…``), quadrature coefficients ``c`` and stencil
coefficients ``d_argK``.  At 10⁶ supports an array-of-records is exactly the
re-read traffic the device path must avoid, so :class:`Items` keeps the same
information *structurally*:

* an item box ``dims`` (first coordinate fastest — the order
  ``Iterators.product`` yields at ``transform.jl:445``),
* integer fields that are affine in the item coordinates (``group_idx = 1 + k``)
  or gathered from an explicit int64 column,
* float fields gathered from a (shared, de-duplicated) float64 array.

``records()`` re-materialises the reference's list-of-NamedTuples for small cases
(tests compare against it).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

MAX_DIMS = 3


@dataclass(frozen=True)
class Field:
    """value(k) = base + Σ step[d]·k_d            (mode 'affine', integer fields)
    value(k) = arr[base + Σ step[d]·k_d]          (mode 'gather')"""

    kind: str  # 'int' | 'float'
    mode: str  # 'affine' | 'gather'
    base: int
    steps: Tuple[int, ...]
    arr: Optional[np.ndarray] = None

    def values(self, dims: Tuple[int, ...]) -> np.ndarray:
        """All item values in item order (first coordinate fastest)."""
        idx = np.full((), self.base, dtype=np.int64)
        for d, n in enumerate(dims):
            shape = [1] * len(dims)
            shape[len(dims) - 1 - d] = n
            idx = idx + (self.steps[d] * np.arange(n, dtype=np.int64)).reshape(shape)
        idx = np.broadcast_to(idx, tuple(reversed(dims))).reshape(-1)
        return idx if self.mode == "affine" else self.arr[idx]

    def _pad(self, before: int, after: int) -> "Field":
        return Field(self.kind, self.mode, self.base, (0,) * before + self.steps + (0,) * after, self.arr)


class Items:
    """Structured item iterator (see module docstring)."""

    def __init__(self, dims: Sequence[int], fields: Dict[str, Field],
                 grid: Optional[Tuple[Tuple[int, ...], Tuple[int, ...]]] = None):
        self.dims = tuple(int(n) for n in dims)
        assert 1 <= len(self.dims) <= MAX_DIMS
        self.fields = dict(fields)
        for f in self.fields.values():
            assert len(f.steps) == len(self.dims)
        # fusion hint: (group ids per dim, grid origin per dim); None = not on a support grid
        self.grid = grid

    # ---- constructors --------------------------------------------------
    @staticmethod
    def single() -> "Items":
        """``[(;)]`` — the one-item iterator of a finite template (transform.jl:440)."""
        return Items((1,), {}, grid=((), ()))

    @staticmethod
    def from_supports(index_name: str, n: int, values: Dict[str, np.ndarray],
                      group_id: Optional[int] = None) -> "Items":
        """Base iterator of one infinite-parameter group (transform.jl:31):
        ``(group_idx = i, alias = support value…)`` for ``i = 1..n``."""
        fields = {index_name: Field("int", "affine", 1, (1,))}
        for name, arr in values.items():
            arr = np.ascontiguousarray(arr, dtype=np.float64)
            assert arr.shape == (n,)
            fields[name] = Field("float", "gather", 0, (1,), arr)
        grid = ((group_id,), (0,)) if group_id is not None else None
        return Items((n,), fields, grid=grid)

    @staticmethod
    def from_records(records: Sequence[dict]) -> "Items":
        """Explicit list of NamedTuple-like dicts (any iterator the structured
        constructors cannot express, e.g. after a domain-restriction filter).
        Integer columns that form an arithmetic progression become affine fields."""
        n = len(records)
        if n == 0:
            raise ValueError("empty item iterator")
        fields: Dict[str, Field] = {}
        for name in records[0].keys():
            col = [r[name] for r in records]
            if all(isinstance(v, (int, np.integer)) and not isinstance(v, bool) for v in col):
                a = np.asarray(col, dtype=np.int64)
                step = int(a[1] - a[0]) if n > 1 else 0
                if n == 1 or np.all(np.diff(a) == step):
                    fields[name] = Field("int", "affine", int(a[0]), (step,))
                else:
                    fields[name] = Field("int", "gather", 0, (1,), a)
            else:
                fields[name] = Field("float", "gather", 0, (1,), np.asarray(col, dtype=np.float64))
        return Items((n,), fields)

    # ---- combinators -----------------------------------------------------
    def __len__(self) -> int:
        return int(np.prod(self.dims))

    def product(self, other: "Items") -> "Items":
        """``vec([merge(i...) for i in Iterators.product(self, other)])`` — self's
        coordinate runs fastest; on a field-name clash ``other`` wins (``merge``)."""
        na, nb = len(self.dims), len(other.dims)
        dims = self.dims + other.dims
        if len(dims) > MAX_DIMS:
            raise ValueError("more than 3 item dimensions")
        fields = {k: f._pad(0, nb) for k, f in self.fields.items()}
        fields.update({k: f._pad(na, 0) for k, f in other.fields.items()})
        grid = None
        if self.grid is not None and other.grid is not None:
            grid = (self.grid[0] + other.grid[0], self.grid[1] + other.grid[1])
        return Items(dims, fields, grid)

    def select(self, start: int, count: int) -> "Items":
        """Contiguous 0-based sub-range of a 1-D iterator (``srt_itr[idxs]`` at
        transform.jl:538 when ``idxs`` is a range)."""
        assert len(self.dims) == 1 and 0 <= start and start + count <= self.dims[0]
        fields = {k: Field(f.kind, f.mode, f.base + f.steps[0] * start, f.steps, f.arr)
                  for k, f in self.fields.items()}
        grid = None
        if self.grid is not None and self.grid[0]:
            grid = (self.grid[0], (self.grid[1][0] + start,))
        return Items((count,), fields, grid)

    def take(self, idxs: Sequence[int]) -> "Items":
        """Arbitrary 0-based subset of a 1-D iterator (contiguous → :meth:`select`)."""
        idxs = np.asarray(idxs, dtype=np.int64)
        if len(idxs) and np.all(np.diff(idxs) == 1):
            return self.select(int(idxs[0]), len(idxs))
        assert len(self.dims) == 1
        fields = {}
        for k, f in self.fields.items():
            vals = f.values(self.dims)[idxs]
            if f.kind == "int":
                fields[k] = Field("int", "gather", 0, (1,), np.ascontiguousarray(vals, dtype=np.int64))
            else:
                fields[k] = Field("float", "gather", 0, (1,), np.ascontiguousarray(vals, dtype=np.float64))
        return Items((len(idxs),), fields)

    def filter(self, mask: np.ndarray) -> "Items":
        """Keep items where ``mask`` (item order) is true (transform.jl:448-451)."""
        mask = np.asarray(mask, dtype=bool).reshape(-1)
        assert mask.shape[0] == len(self)
        flat = self.flatten()
        return flat.take(np.nonzero(mask)[0])

    def flatten(self) -> "Items":
        """Same items as a 1-D iterator with explicit columns where needed."""
        if len(self.dims) == 1:
            return self
        n = len(self)
        fields = {}
        for k, f in self.fields.items():
            vals = np.ascontiguousarray(f.values(self.dims))
            fields[k] = Field(f.kind, "gather", 0, (1,), vals.astype(np.int64 if f.kind == "int" else np.float64))
        return Items((n,), fields)

    def with_float(self, name: str, values: np.ndarray) -> "Items":
        """Add one Float64 per item of a 1-D iterator (``c``, ``d_argK``)."""
        assert len(self.dims) == 1
        arr = np.ascontiguousarray(values, dtype=np.float64)
        assert arr.shape == (self.dims[0],)
        fields = dict(self.fields)
        fields[name] = Field("float", "gather", 0, (1,), arr)
        return Items(self.dims, fields, self.grid)

    def with_int_affine(self, name: str, base: int, step: int) -> "Items":
        assert len(self.dims) == 1
        fields = dict(self.fields)
        fields[name] = Field("int", "affine", int(base), (int(step),))
        return Items(self.dims, fields, self.grid)

    def scaled_float(self, name: str, factor_field_of_other: np.ndarray) -> "Items":  # pragma: no cover
        raise NotImplementedError

    # ---- materialisation ---------------------------------------------------
    def column(self, name: str) -> np.ndarray:
        return self.fields[name].values(self.dims)

    def records(self) -> List[dict]:
        cols = {k: self.column(k) for k in self.fields}
        out = []
        for k in range(len(self)):
            out.append({name: (int(c[k]) if self.fields[name].kind == "int" else float(c[k]))
                        for name, c in cols.items()})
        return out


def as_items(itr) -> Items:
    if isinstance(itr, Items):
        return itr
    recs = list(itr)
    if len(recs) == 1 and len(recs[0]) == 0:
        return Items.single()
    return Items.from_records(recs)
