"""Chain KKT solver: block cyclic reduction over the time-banded augmented system, on the device.

The reference hands the linear solve that follows ``jac_coord!/hess_coord!`` in every interior-point iteration to
MadNLPGPU + CUDSS (``/root/reference/README.md:36-37``).  ROCm 7.2 has no CUDSS; rocSOLVER's sparse re-factorisation
(``kkt.py``) is correct but sees the time-banded KKT matrix as one dependency chain (49 ms at 2 000 supports, 4.2 s at
10⁵).  The structure a transcription gives the matrix is far more specific:

* supports of the derivative's parameter couple ONLY through the stencil (``src/transform.jl:535-557``: ``x_k[i-1]`` for
  backward differences; an element's first node for orthogonal collocation, ``:581-584``), so with the unknowns grouped
  by support — block ``k`` = the variables AND the constraint rows (their multipliers) of support(s) ``k``, variables
  first — ``K`` is block tridiagonal;
* finite / first-stage variables (``transform.jl:104-117``) and the rows that touch nothing else couple to every block:
  a small dense border.

:class:`ChainLayout` derives that grouping from the model's slab table (``core.slabs``) and the Jacobian structure — no
model-specific code — and :class:`ChainKKT` keeps the dense blocks ``D | B | E | G`` on the device, fills them from the
CSR values ``kkt.KKTSystem.assemble`` produces (one ``index_copy_``), and calls the hand-written kernels behind
``iem_kkt_chain_factor / iem_kkt_chain_solve`` (``csrc/iem_kkt_device.h``: Gauss-Jordan inverses of the pivot blocks in
LDS, ⌈log₂ S⌉ levels of block cyclic reduction, per-block border Schur terms, pivot signs counted for the inertia).
Two-dimensional models whose blocks would be too large (pandemic at 100 scenarios: 1 701 unknowns per time support) are
refused with a clear message — :class:`kkt.KKTSystem` remains the general path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import lib as _lib

MAX_NB, MAX_NE = 96, 64


def fits_lds(nb: int, ne: int) -> bool:
    """two nb x nb tiles + two nb x ne tiles of doubles (odd row strides) in the 160 KB of a CU (csrc/iem_kkt_device.h)"""
    return 16 * nb * (nb + 1) + 16 * nb * (ne + 1) + 4096 <= 160 * 1024


def _ceil4(n: int) -> int:
    return (int(n) + 3) // 4 * 4


class ChainLayout:
    """Grouping of the ``nvar + ncon`` KKT unknowns into ``S`` chain blocks of ``nb`` (padded) unknowns and a border of
    ``ne``; unknown ``u < nvar`` is variable ``u``, ``u >= nvar`` the multiplier of row ``u - nvar``.

    ``blk[u]`` = block (``-1``: border), ``loc[u]`` = position inside the block / the border; ``counts[k]`` = real
    unknowns of block ``k``.  ``group``: the infinite-parameter group the chain runs along (default: the one whose
    supports a constraint row couples, else the one with the most supports)."""

    def __init__(self, slabs, nvar: int, ncon: int, jac_rows, jac_cols, group: Optional[int] = None,
                 max_nb: int = MAX_NB, max_ne: int = MAX_NE):
        import torch
        jr = torch.as_tensor(np.asarray(jac_rows, dtype=np.int64))
        jc = torch.as_tensor(np.asarray(jac_cols, dtype=np.int64))
        self.nvar, self.ncon = int(nvar), int(ncon)
        groups = sorted({g for _, _, gs in slabs for g in gs if g > 0})
        if not groups:
            raise _lib.IemError("chain KKT: the model has no infinite-parameter slab table (nothing to chain along)")

        def coords(g):
            vc = np.full(self.nvar, -1, dtype=np.int64)
            for off, dims, gs in slabs:
                if g in gs:
                    a = list(gs).index(g)
                    n = int(np.prod(dims))
                    stride = int(np.prod(dims[:a])) if a else 1
                    vc[off:off + n] = (np.arange(n) // stride) % dims[a]
            return vc

        def row_span(vc):
            v = torch.as_tensor(vc)[jc]
            hi = torch.full((self.ncon,), -1, dtype=torch.int64).scatter_reduce(0, jr, v, "amax")
            big = torch.iinfo(torch.int64).max
            lo = torch.full((self.ncon,), big, dtype=torch.int64).scatter_reduce(0, jr, torch.where(v >= 0, v, torch.tensor(big)), "amin")
            lo = torch.where(lo == big, hi, lo)
            return hi.numpy(), lo.numpy()

        if group is None:
            best = None
            for g in groups:
                vc = coords(g)
                hi, lo = row_span(vc)
                reach = int((hi - lo).max()) if self.ncon else 0
                key = (reach > 0, int(vc.max()) + 1)
                if best is None or key > best[0]:
                    best = (key, g, vc, hi, lo)
            _, group, vc, hi, lo = best
        else:
            vc = coords(group)
            hi, lo = row_span(vc)
        self.group = int(group)
        self.reach = int((hi - lo).max()) if self.ncon else 0
        R = max(self.reach, 1)                      # supports per block: a row then spans at most two consecutive blocks
        self.supports_per_block = R
        chain = np.concatenate([vc, hi])            # variables, then rows (a row sits with the LAST support it touches)
        blk = np.where(chain >= 0, chain // R, -1)
        self.S = int(blk.max()) + 1 if (blk >= 0).any() else 0
        if self.S < 1:
            raise _lib.IemError("chain KKT: no unknown lies on the chain")
        n = self.nvar + self.ncon
        kind = (np.arange(n) >= self.nvar).astype(np.int64)
        on = blk >= 0
        ids = np.nonzero(on)[0]
        order = ids[np.lexsort((ids, kind[ids], blk[ids]))]          # by block, variables first, then by index
        counts = np.bincount(blk[ids], minlength=self.S)
        start = np.concatenate([[0], np.cumsum(counts)])
        loc = np.full(n, -1, dtype=np.int64)
        loc[order] = np.arange(order.size) - start[blk[order]]
        border = np.nonzero(~on)[0]                                  # border: variables first, then rows (already in that order)
        loc[border] = np.arange(border.size)
        self.blk, self.loc, self.counts = blk, loc, counts
        self.n_border = int(border.size)
        self.nb, self.ne = _ceil4(counts.max()), _ceil4(border.size)
        if self.nb > max_nb or self.ne > max_ne or not fits_lds(self.nb, self.ne):
            raise _lib.IemError(f"chain KKT: blocks of {int(counts.max())} unknowns / a border of {border.size} exceed the dense-block solver's "
                                f"limits ({max_nb} / {max_ne}, two tiles of each in LDS); use kkt.KKTSystem (rocSOLVER re-factorisation) for this model")

    # offsets of D | B | E | G in the flat block buffer
    def offsets(self):
        S, nb, ne = self.S, self.nb, self.ne
        oD, oB = 0, S * nb * nb
        oE = oB + S * nb * nb
        oG = oE + S * nb * ne
        return oD, oB, oE, oG, oG + ne * ne

    def scatter_plan(self, rows, cols):
        """For the entries ``(rows[i], cols[i])`` of K (both triangles): ``(src, dest)`` — entry ``src[i]`` goes to flat
        position ``dest[i]``; the upper coupling blocks and the border's row block are dropped (symmetry)."""
        rows, cols = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
        kr, kc, lr, lc = self.blk[rows], self.blk[cols], self.loc[rows], self.loc[cols]
        oD, oB, oE, oG, _ = self.offsets()
        nb, ne = self.nb, self.ne
        dest = np.full(rows.size, -1, dtype=np.int64)
        both = (kr >= 0) & (kc >= 0)
        diag = both & (kr == kc)
        low = both & (kr == kc + 1)
        bad = both & (np.abs(kr - kc) > 1)
        if bad.any():
            raise _lib.IemError("chain KKT: an entry couples blocks that are not neighbours (the chain grouping does not fit this model)")
        dest[diag] = oD + (kr[diag] * nb + lr[diag]) * nb + lc[diag]
        dest[low] = oB + (kr[low] * nb + lr[low]) * nb + lc[low]
        e = (kr >= 0) & (kc < 0)
        dest[e] = oE + (kr[e] * nb + lr[e]) * ne + lc[e]
        g = (kr < 0) & (kc < 0)
        dest[g] = oG + lr[g] * ne + lc[g]
        src = np.nonzero(dest >= 0)[0]
        return src, dest[src]

    def pad_positions(self):
        """Flat positions of the unit diagonal of the padding (blocks shorter than ``nb``, border shorter than ``ne``)."""
        oD, _, _, oG, _ = self.offsets()
        nb, ne = self.nb, self.ne
        k = np.repeat(np.arange(self.S), nb - self.counts)
        l = np.concatenate([np.arange(c, nb) for c in self.counts]) if (self.counts < nb).any() else np.zeros(0, np.int64)
        pd = oD + (k * nb + l) * nb + l
        lb = np.arange(self.n_border, ne)
        return np.concatenate([pd, oG + lb * ne + lb]).astype(np.int64)

    def positions(self):
        """``(chain unknowns, their positions in the S·nb block vector, border unknowns)``."""
        on = np.nonzero(self.blk >= 0)[0]
        return on, self.blk[on] * self.nb + self.loc[on], np.nonzero(self.blk < 0)[0]


class ChainKKT:
    """Factor / solve the augmented system assembled by a :class:`kkt.KKTSystem` with the chain solver."""

    def __init__(self, kkt, group: Optional[int] = None):
        import torch
        self._torch = torch
        self.kkt = kkt
        m = self.model = kkt.model
        if m.core is None:
            raise _lib.IemError("chain KKT needs the model's core (slab table)")
        jr, jc = m.jac_structure(0)
        self.layout = L = ChainLayout(m.core.slabs, m.meta.nvar, m.meta.ncon, jr, jc, group=group)
        dev = m.device
        rowptr = kkt.rowptr.cpu().numpy().astype(np.int64)
        rows = np.repeat(np.arange(kkt.n), np.diff(rowptr))
        src, dest = L.scatter_plan(rows, kkt.colind.cpu().numpy())
        self._src, self._dest = torch.as_tensor(src, device=dev), torch.as_tensor(dest, device=dev)
        self._pad = torch.as_tensor(L.pad_positions(), device=dev)
        oD, oB, oE, oG, total = L.offsets()
        S, nb, ne = L.S, L.nb, L.ne
        f64 = dict(dtype=torch.float64, device=dev)
        self.flat = torch.zeros(total, **f64)
        self.D, self.B = self.flat[oD:oB], self.flat[oB:oE]
        self.E, self.G = self.flat[oE:oG], self.flat[oG:total].view(ne, ne)
        nxy = S * nb * nb if L.reach > 0 else 1
        self.X, self.Y = torch.empty(nxy, **f64), torch.empty(nxy, **f64)
        self.Z, self.Gp = torch.empty(max(S * nb * ne, 1), **f64), torch.empty(max(S * ne * ne, 1), **f64)
        self.info = torch.zeros(3, dtype=torch.int64, device=dev)
        self._r = torch.zeros(S * nb, **f64)
        self._rBp = torch.empty(max(S * ne, 1), **f64)
        on, pos, border = L.positions()
        self._on, self._pos, self._border = (torch.as_tensor(a, device=dev) for a in (on, pos, border))
        self._glu = None
        self.negative_pivots = None

    def load(self):
        """Dense blocks from the CSR values of the last ``kkt.assemble`` (zero fill + one scatter)."""
        self.flat.zero_()
        if self._pad.numel():
            self.flat[self._pad] = 1.0
        self.flat.index_copy_(0, self._dest, self.kkt.vals[self._src])
        return self

    def factor(self, tiny: float = 1e-30):
        """Block cyclic reduction in place; afterwards ``inertia()`` and ``solve()``."""
        t, L, m = self._torch, self.layout, self.model
        m._sync_stream()
        p = lambda a: C.c_void_p(a.data_ptr())
        chained = L.reach > 0      # reach 0 (scenario blocks of a two-stage problem): one launch, no levels
        _lib.check(m._L.iem_kkt_chain_factor(m._h, L.S, L.nb, L.ne, p(self.D), p(self.B) if chained else None, p(self.X) if chained else None,
                                            p(self.Y) if chained else None, p(self.E), p(self.Z), p(self.Gp), p(self.info), float(tiny)))
        if L.ne:
            Gs = self.G - self.Gp[:L.S * L.ne * L.ne].view(L.S, L.ne, L.ne).sum(0)
            self._Gs = Gs
            self._glu = t.linalg.lu_factor(Gs)
        return self

    def inertia(self):
        """``(positive, negative, doubtful)`` pivots of the whole system (padding excluded); a correctly regularised KKT
        matrix has exactly ``ncon`` negative ones.  Synchronises."""
        t, L = self._torch, self.layout
        info = self.info.cpu().numpy()
        neg = int(info[0])
        if L.ne:
            ev = t.linalg.eigvalsh(self._Gs)
            neg += int((ev < 0).sum().item())
        n = self.layout.nvar + self.layout.ncon
        return n - neg, neg, int(info[1])

    def solve(self, rhs, refine: int = 1):
        """``K x = rhs`` (device tensor of length ``nvar + ncon``) with the current factors; ``refine`` steps of iterative
        refinement against the CSR matrix."""
        t = self._torch
        x = self._solve_once(rhs)
        for _ in range(refine):
            res = rhs - self._matvec(x)
            x = x + self._solve_once(res)
        return x

    def _matvec(self, x):
        """``K x`` with the CSR matrix of the last ``kkt.assemble`` (``iem_csr_spmv``)."""
        k, m = self.kkt, self.model
        y = self._torch.empty_like(x)
        m._sync_stream()
        p = lambda a: C.c_void_p(a.data_ptr())
        if getattr(self, "_long_rows", None) is None:      # rows beyond IEM_SPMV_LONG_ROW entries get a workgroup each
            self._long_rows = self._torch.nonzero((k.rowptr[1:] - k.rowptr[:-1]) > 256).flatten().to(self._torch.int64)
        nl = int(self._long_rows.numel())
        _lib.check(m._L.iem_csr_spmv(m._h, k.n, p(k.rowptr), p(k.colind), p(k.vals), p(x), p(y), nl, p(self._long_rows) if nl else None))
        return y

    def _solve_once(self, rhs):
        t, L, m = self._torch, self.layout, self.model
        r = self._r
        r.zero_()
        r[self._pos] = rhs[self._on]
        m._sync_stream()
        p = lambda a: C.c_void_p(a.data_ptr())
        chained = L.reach > 0
        args = (m._h, L.S, L.nb, L.ne, p(self.D), p(self.X) if chained else None, p(self.Y) if chained else None, p(self.Z), p(r), p(self._rBp))
        xB = None
        _lib.check(m._L.iem_kkt_chain_solve(*args, None, 0))
        if L.ne:
            rB = t.zeros(L.ne, dtype=t.float64, device=r.device)
            rB[:L.n_border] = rhs[self._border]
            rB = rB - self._rBp[:L.S * L.ne].view(L.S, L.ne).sum(0)
            xB = t.linalg.lu_solve(*self._glu, rB.unsqueeze(1)).squeeze(1).contiguous()
        _lib.check(m._L.iem_kkt_chain_solve(*args, p(xB) if xB is not None else None, 1))
        out = t.empty_like(rhs)
        out[self._on] = r[self._pos]
        if L.ne:
            out[self._border] = xB[:L.n_border]
        return out
