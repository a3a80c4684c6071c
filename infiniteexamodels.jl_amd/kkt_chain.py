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
model-specific code — and :class:`ChainKKT` keeps the blocks ``D | Bt | E | G`` on the device, fills them from the
CSR values ``kkt.KKTSystem.assemble`` produces (one ``index_copy_``), and calls the hand-written kernels behind
``iem_kkt_chain_factor / iem_kkt_chain_solve`` (``csrc/iem_kkt_device.h``: block Gauss-Jordan inverses of the pivot blocks
on the FP64 matrix cores, ⌈log₂ S⌉ levels of block cyclic reduction over NARROW couplings — only the rows / columns the
stencil touches —, per-block border Schur terms, pivot signs counted for the inertia).
Two-dimensional models whose blocks would be too large (pandemic at 100 scenarios: 1 701 unknowns per time support) are
refused with a clear message — :class:`kkt.KKTSystem` remains the general path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import lib as _lib

MAX_NB, MAX_NE, MAX_NC = 96, 128, 48


def fits_lds(nb: int, ne: int, nc: int = 4) -> bool:
    """LDS of the kernels (csrc/iem_kkt_device.h): the panel buffers of the inverse and, with a border, one nb x nb and two
    nb x ne tiles of doubles (odd row strides); seven nc x nc tiles — each kernel within the 160 KB of a CU."""
    elim = 8 * (10 * nb + 8 * (nb + 1)) + (8 * nb * (nb + 1) + 16 * nb * (ne + 1) if ne else 0) + 1024
    return elim <= 160 * 1024 and 56 * nc * (nc + 1) + 1024 <= 160 * 1024


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
        # LANES (2-D support grids, ESCAPE34/pandemic.jl: t x xi): lane of a variable = its flattened position along the OTHER
        # dimensions of a slab that carries the chain's group (-1: none).  When the plain chain's blocks are too large, every
        # lane gets a chain of its own (block = lane * blocks-per-lane + time block, zero coupling at the seams) and the
        # variables on the chain's parameter alone (u(t)) go to the border — csrc/iem_kkt_host.hpp does the same.
        vlane = np.full(self.nvar, -1, dtype=np.int64)
        nlanes, consistent = 1, True
        for off, dims, gs in slabs:
            if group in gs:
                a = list(gs).index(group)
                n_ = int(np.prod(dims))
                stride = int(np.prod(dims[:a])) if a else 1
                other = n_ // dims[a]
                if other > 1:
                    consistent = consistent and nlanes in (1, other)
                    nlanes = max(nlanes, other)
                    i = np.arange(n_)
                    vlane[off:off + n_] = (i // (stride * dims[a])) * stride + (i % stride)
        jrn, jcn = jr.numpy(), jc.numpy()
        n = self.nvar + self.ncon
        kind = (np.arange(n) >= self.nvar).astype(np.int64)

        def build(use_lanes: bool):
            if use_lanes:
                laned = (vlane[jcn] >= 0) & (vc[jcn] >= 0)
                rl = np.full(self.ncon, -1, dtype=np.int64)
                rhi = np.full(self.ncon, -1, dtype=np.int64)
                rlo = np.full(self.ncon, np.iinfo(np.int64).max, dtype=np.int64)
                np.maximum.at(rl, jrn[laned], vlane[jcn[laned]])
                rmin = np.full(self.ncon, np.iinfo(np.int64).max, dtype=np.int64)
                np.minimum.at(rmin, jrn[laned], vlane[jcn[laned]])
                if ((rl >= 0) & (rmin != rl)).any():
                    raise _lib.IemError("chain KKT: a constraint row couples two lanes of the support grid")
                np.maximum.at(rhi, jrn[laned], vc[jcn[laned]])
                np.minimum.at(rlo, jrn[laned], vc[jcn[laned]])
                vch = np.where((vlane >= 0) & (vc >= 0), vc, -1)
                chain = np.concatenate([vch, np.where(rl >= 0, rhi, -1)])
                lane = np.concatenate([np.where(vch >= 0, vlane, 0), np.where(rl >= 0, rl, 0)])
                lanes = nlanes
                span = np.where(rhi >= 0, rhi - np.where(rlo == np.iinfo(np.int64).max, rhi, rlo), 0)
            else:
                chain = np.concatenate([vc, hi])            # variables, then rows (a row sits with the LAST support it touches)
                lane = np.zeros(n, dtype=np.int64)
                lanes = 1
                span = hi - lo
            reach = int(span.max()) if self.ncon else 0
            R = max(reach, 1)                       # supports per block: a row then spans at most two consecutive blocks
            on = chain >= 0
            ids = np.nonzero(on)[0]
            border = np.nonzero(~on)[0]                 # border: variables first, then rows (already in that order)

            def arrange(phase):
                """Blocks of R supports starting at support -phase: (blk, loc, counts, nb, coupling rows, coupling columns).
                Inside a block the unknowns sit in FIXED places — variables before rows, each kind by the support's position in
                the block, then by index, every (kind, position) group at the same base in every block — so that a block with
                fewer unknowns (the first support has no difference row, the last block may be short) leaves holes instead of
                shifting the others: the coupling then lives on the same few local rows / columns in every block."""
                tb = np.where(on, (chain + phase) // R, -1)
                Sb = int(tb.max()) + 1 if on.any() else 0
                blk = np.where(on, lane * Sb + tb, -1)
                off = np.where(on, (chain + phase) % R, 0)
                S = lanes * Sb
                gk = (blk[ids] * 2 + kind[ids]) * R + off[ids]       # (block, kind, position in the block); ids ascend: a stable sort keeps index order
                perm = np.argsort(gk, kind="stable")
                order, g = ids[perm], gk[perm]
                first = np.concatenate([[True], g[1:] != g[:-1]]) if order.size else np.zeros(0, bool)
                pos = np.arange(order.size)
                ordinal = pos - np.maximum.accumulate(np.where(first, pos, 0))
                size = np.bincount(g, minlength=S * 2 * R).reshape(S, 2, R).max(axis=0) if S else np.zeros((2, R), dtype=np.int64)   # largest (kind, position) group over the blocks
                base = np.zeros((2, R), dtype=np.int64)
                base[0] = np.concatenate([[0], np.cumsum(size[0])[:-1]])
                base[1] = size[0].sum() + np.concatenate([[0], np.cumsum(size[1])[:-1]])
                loc = np.full(n, -1, dtype=np.int64)
                loc[order] = base[kind[order], off[order]] + ordinal
                loc[border] = np.arange(border.size)
                counts = np.bincount(blk[ids], minlength=S)
                nb = _ceil4(size.sum())
                # the coupling K[block k, block k-1]: J entries whose row sits one block after the variable, and the transposes
                # of those whose variable sits one block after the row
                kr, kc = blk[self.nvar + jrn], blk[jcn]
                low = (kr >= 0) & (kc >= 0) & (kr == kc + 1)
                up = (kr >= 0) & (kc >= 0) & (kc == kr + 1)
                rows = np.unique(np.concatenate([loc[self.nvar + jrn[low]], loc[jcn[up]]]))
                cols = np.unique(np.concatenate([loc[jcn[low]], loc[self.nvar + jrn[up]]]))
                return blk, loc, counts, nb, rows, cols, S

            best = None
            for phase in range(R):                      # where the blocks start decides how WIDE the coupling is (collocation: blocks that
                cand = arrange(phase)                   # end on an element boundary couple through the boundary node only)
                key = (max(cand[4].size, cand[5].size), cand[3], phase)
                if best is None or key < best[0]:
                    best = (key, cand)
            blk, loc, counts, nb, rows, cols, S = best[1]
            if S < 1:
                raise _lib.IemError("chain KKT: no unknown lies on the chain")
            ne = _ceil4(border.size)
            if nb > max_nb or ne > max_ne or not fits_lds(nb, ne):
                raise _lib.IemError(f"chain KKT: blocks of {int(counts.max())} unknowns / a border of {border.size} exceed the dense-block solver's "
                                    f"limits ({max_nb} / {max_ne}, the tiles of a block in LDS); use kkt.KKTSystem (rocSOLVER re-factorisation) for this model")
            return dict(reach=reach, R=R, blk=blk, loc=loc, counts=counts, nb=nb, ne=ne, rows=rows, cols=cols, S=S, phase=best[0][2],
                        n_border=int(border.size), lanes=lanes)

        try:                                        # the plain chain first (every model it fits keeps its layout)
            c = build(False)
        except _lib.IemError as plain:
            if nlanes <= 1 or not consistent:
                raise
            try:
                c = build(True)
            except _lib.IemError as laned:
                raise _lib.IemError(f"{plain}; one chain per lane of the support grid ({nlanes} lanes): {laned}") from None
        self.reach, self.supports_per_block, self.lanes = c["reach"], c["R"], c["lanes"]
        self.blk, self.loc, self.counts, self.S, self.phase = c["blk"], c["loc"], c["counts"], c["S"], c["phase"]
        self.n_border, self.nb, self.ne = c["n_border"], c["nb"], c["ne"]
        rows, cols = c["rows"], c["cols"]
        # The coupling is NARROW: entries on a few local rows R of block k (the derivative-approximation rows of its first
        # support) and a few local columns C of block k-1 (the differentiated states) — from the Jacobian's structure here;
        # set_coupling() widens it if the caller's K holds more (a Hessian entry across two supports).
        self.set_coupling(rows, cols)

    def set_coupling(self, rows, cols):
        """Local rows (of block k) and columns (of block k-1) the coupling blocks live on; ``nc`` = their padded count."""
        self.rowsR, self.colsC = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
        self.nc = max(_ceil4(max(self.rowsR.size, self.colsC.size)), 4)
        if self.reach > 0 and (self.nc > MAX_NC or self.nc > self.nb or not fits_lds(self.nb, self.ne, self.nc)):
            raise _lib.IemError(f"chain KKT: the coupling between neighbouring blocks spans {self.rowsR.size} rows / {self.colsC.size} columns "
                                f"(limit {MAX_NC}); use kkt.KKTSystem (rocSOLVER re-factorisation) for this model")
        self._ridx = np.full(self.nb, -1, dtype=np.int64); self._ridx[self.rowsR] = np.arange(self.rowsR.size)
        self._cidx = np.full(self.nb, -1, dtype=np.int64); self._cidx[self.colsC] = np.arange(self.colsC.size)

    def coupling_tables(self):
        """``(rows, cols)`` as int32 arrays of length ``nc`` (-1 = padding) for the kernels."""
        r = np.full(self.nc, -1, dtype=np.int32); r[:self.rowsR.size] = self.rowsR
        c = np.full(self.nc, -1, dtype=np.int32); c[:self.colsC.size] = self.colsC
        return r, c

    # offsets of D | Bt | E | G in the flat block buffer
    def offsets(self):
        S, nb, ne, nc = self.S, self.nb, self.ne, self.nc
        oD, oB = 0, S * nb * nb
        oE = oB + S * nc * nc
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
        if low.any():
            ri, ci = self._ridx[lr[low]], self._cidx[lc[low]]
            if (ri < 0).any() or (ci < 0).any():       # K couples more than the Jacobian showed: widen R / C and start over
                self.set_coupling(np.union1d(self.rowsR, lr[low]), np.union1d(self.colsC, lc[low]))
                return self.scatter_plan(rows, cols)
            dest[low] = oB + (kr[low] * self.nc + ri) * self.nc + ci
        e = (kr >= 0) & (kc < 0)
        dest[e] = oE + (kr[e] * nb + lr[e]) * ne + lc[e]
        g = (kr < 0) & (kc < 0)
        dest[g] = oG + lr[g] * ne + lc[g]
        src = np.nonzero(dest >= 0)[0]
        return src, dest[src]

    def pad_positions(self):
        """Flat positions of the unit diagonal of the padding (the places of a block no unknown occupies, the border's tail)."""
        oD, _, _, oG, _ = self.offsets()
        nb, ne = self.nb, self.ne
        free = np.ones(self.S * nb, dtype=bool)
        on = self.blk >= 0
        free[self.blk[on] * nb + self.loc[on]] = False
        slot = np.nonzero(free)[0]
        k, l = slot // nb, slot % nb
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
        oD, oB, oE, oG, total = L.offsets()          # (after scatter_plan: it may have widened the coupling)
        S, nb, ne, nc = L.S, L.nb, L.ne, L.nc
        f64 = dict(dtype=torch.float64, device=dev)
        self.flat = torch.zeros(total, **f64)
        self.D, self.B = self.flat[oD:oB], self.flat[oB:oE]
        self.E, self.G = self.flat[oE:oG], self.flat[oG:total].view(ne, ne)
        self.BR = torch.zeros(S * nc * nc if L.reach > 0 else 1, **f64)
        rt, ct = L.coupling_tables()
        self._rows, self._cols = torch.as_tensor(rt, device=dev), torch.as_tensor(ct, device=dev)
        self.Z, self.Gp = torch.empty(max(S * nb * ne, 1), **f64), torch.empty(max(S * ne * ne, 1), **f64)
        self.info = torch.zeros(3, dtype=torch.int64, device=dev)
        self._r = torch.zeros(S * nb, **f64)
        self._z = torch.empty(S * nb if L.reach > 0 else 1, **f64)
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
        _lib.check(m._L.iem_kkt_chain_factor(m._h, L.S, L.nb, L.ne, L.nc, p(self.D), p(self.B) if chained else None, p(self.BR) if chained else None,
                                            p(self._rows) if chained else None, p(self._cols) if chained else None, p(self.E), p(self.Z), p(self.Gp),
                                            p(self.info), float(tiny)))
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

    def solve(self, rhs, refine=1, rtol: float = 1e-9):
        """``K x = rhs`` (device tensor of length ``nvar + ncon``) with the current factors; ``refine`` steps of iterative
        refinement against the CSR matrix — or ``refine = "auto"``: up to two steps, each only while the residual exceeds
        ``rtol * max|rhs|`` (one product with the CSR matrix decides; a solve is eight times that)."""
        x = self._solve_once(rhs)
        if refine == "auto":
            bound = rtol * max(1.0, float(rhs.abs().max().item()))
            for _ in range(2):
                res = rhs - self._matvec(x)
                if float(res.abs().max().item()) <= bound:
                    break
                x = x + self._solve_once(res)
            return x
        for _ in range(int(refine)):
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
        args = (m._h, L.S, L.nb, L.ne, L.nc, p(self.D), p(self.B) if chained else None, p(self.BR) if chained else None,
                p(self._rows) if chained else None, p(self._cols) if chained else None, p(self.Z), p(r), p(self._z) if chained else None, p(self._rBp))
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
