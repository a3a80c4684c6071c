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
                 max_nb: int = MAX_NB, max_ne: int = MAX_NE, hubs: bool = False):
        """``hubs``: the layout :class:`HubChainKKT` works on — always one chain per lane (reduced lane by lane:
        ``iem_kkt_chain_level``'s ``lane_len``), and NO limit on the border (the chain parameter's own variables, ``u(t)``: handled as span-sparse
        border columns outside the chain kernels)."""
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
            if hubs:
                if nb > max_nb or not fits_lds(nb, 0):
                    raise _lib.IemError(f"chain KKT: blocks of {int(counts.max())} unknowns exceed the dense-block solver's limit ({max_nb})")
            elif nb > max_nb or ne > max_ne or not fits_lds(nb, ne):
                raise _lib.IemError(f"chain KKT: blocks of {int(counts.max())} unknowns / a border of {border.size} exceed the dense-block solver's "
                                    f"limits ({max_nb} / {max_ne}, the tiles of a block in LDS); use kkt.KKTSystem (rocSOLVER re-factorisation) for this model")
            ph = best[0][2]
            # time block of every unknown that has a chain coordinate at all (hubs: the border's own place on the time axis)
            allchain = np.concatenate([vc, hi])
            tblock = np.where(allchain >= 0, (allchain + ph) // R, -1)
            return dict(reach=reach, R=R, blk=blk, loc=loc, counts=counts, nb=nb, ne=ne, rows=rows, cols=cols, S=S, phase=ph,
                        n_border=int(border.size), lanes=lanes, tblock=tblock)

        try:                                        # the plain chain first (every model it fits keeps its layout)
            if hubs:
                if nlanes <= 1 or not consistent:
                    raise _lib.IemError("chain KKT (hub border): the chain's parameter has no second dimension to cut lanes along")
                c = build(True)
            else:
                c = build(False)
        except _lib.IemError as plain:
            if hubs:
                raise
            if nlanes <= 1 or not consistent:
                raise
            try:
                c = build(True)
            except _lib.IemError as laned:
                raise _lib.IemError(f"{plain}; one chain per lane of the support grid ({nlanes} lanes): {laned}") from None
        self.reach, self.supports_per_block, self.lanes = c["reach"], c["R"], c["lanes"]
        self.blk, self.loc, self.counts, self.S, self.phase = c["blk"], c["loc"], c["counts"], c["S"], c["phase"]
        self.n_border, self.nb, self.ne = c["n_border"], c["nb"], c["ne"]
        self.tblock = c["tblock"]
        rows, cols = c["rows"], c["cols"]
        # The coupling is NARROW: entries on a few local rows R of block k (the derivative-approximation rows of its first
        # support) and a few local columns C of block k-1 (the differentiated states) — from the Jacobian's structure here;
        # set_coupling() widens it if the caller's K holds more (a Hessian entry across two supports).
        self.set_coupling(rows, cols)

    def set_coupling(self, rows, cols):
        """Local rows (of block k) and columns (of block k-1) the coupling blocks live on; ``nc`` = their padded count."""
        self.rowsR, self.colsC = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
        self.nc = max(_ceil4(max(self.rowsR.size, self.colsC.size)), 4)
        if self.reach > 0 and (self.nc > MAX_NC or self.nc > self.nb or not fits_lds(self.nb, min(self.ne, MAX_NE), self.nc)):
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


class HubChainKKT:
    """The chain solver on a 2-D support grid whose border is too large for the dense-border kernels: BASELINE config 3,
    ``ESCAPE34/pandemic.jl`` at 5 000 x 100 supports — one chain per scenario (lanes), and 5 000 HUBS ``u(t)``, each coupled to
    the blocks of its own time support in every lane.

    Block elimination with a SPAN-SPARSE border.  Eliminating block ``i`` of the chain (cyclic reduction, level ``s``) adds
    ``-B D_i^-1 E_i`` to its neighbours' border columns and ``-E_i' D_i^-1 E_i`` to the hubs' Schur complement ``S``.  A block at
    time ``t`` alive at level ``s`` only ever touches the hubs ``t - (s - 1) .. t + (s - 1)``: ``E`` is kept as ``nb x (2s - 1)``
    per block (time-major, lanes contiguous), widened level by level; the contributions to ``S`` of the blocks eliminated at one
    level fall on DISJOINT diagonal blocks of ``S`` and are batched matrix products over ``lanes x nb`` — the dense products the
    matrix cores are for (rocBLAS through torch: plain library GEMMs).  The chain itself (block inverses on the FP64 matrix
    cores, narrow couplings) runs through the same kernels as every other model, one level at a time
    (``iem_kkt_chain_level``), with no border at all.  ``S`` (hubs x hubs, dense) is factorised by Cholesky: under an
    interior-point regularisation with the correct inertia it is positive definite (the hubs are primal variables); a
    failing Cholesky is reported as doubtful inertia.  Solve: two chain solves around the dense one
    (``x_B = S^-1 (r_B - E' K_c^-1 r_c)``, ``x_c = K_c^-1 (r_c - E x_B)``), ``E`` being the ORIGINAL one-column-per-block border.

    ``levels``: ``None`` = the device kernels; tests pass a dense restatement (``tests/chain_reference.py``) to check the hub
    pipeline itself on CPU tensors."""

    def __init__(self, kkt, group: Optional[int] = None, levels=None, device=None):
        import torch
        self._torch = t = torch
        self.kkt = kkt
        m = self.model = kkt.model
        jr, jc = m.jac_structure(0)
        self.layout = L = ChainLayout(m.core.slabs, m.meta.nvar, m.meta.ncon, jr, jc, group=group, hubs=True)
        if L.reach != 1 or L.supports_per_block != 1:
            raise _lib.IemError("chain KKT (hub border): only stencils of reach 1 (backward differences) are handled")
        dev = self.device = device if device is not None else m.device
        self.lanes, self.S, self.nb = int(L.lanes), int(L.S), int(L.nb)
        self.Tp = Tp = self.S // self.lanes
        n = L.nvar + L.ncon
        border = np.nonzero(L.blk < 0)[0]
        tb = L.tblock[border]
        if (tb < 0).any():
            raise _lib.IemError("chain KKT (hub border): a border unknown has no place on the chain's axis (finite variables are not handled here)")
        order = np.argsort(tb, kind="stable")
        cnt = np.bincount(tb, minlength=Tp)
        self.hw = hw = int(cnt.max())
        first = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        ordinal = np.empty(border.size, dtype=np.int64)
        ordinal[order] = np.arange(border.size) - first[tb[order]]
        self.hub_of = np.full(n, -1, dtype=np.int64)
        self.hub_of[border] = tb * hw + ordinal
        self.T = int(tb.max()) + 1                      # time blocks that own a hub
        self.P = P = 1 << max(Tp - 1, 0).bit_length()   # the spans of the last levels reach up to the next power of two of time blocks
        self.H, self.Hp = self.T * hw, P * hw
        # where the entries of K go: D | Bt through the layout's own plan (it may widen the coupling), E0 and S here
        rowptr = kkt.rowptr.cpu().numpy().astype(np.int64)
        rows = np.repeat(np.arange(kkt.n), np.diff(rowptr))
        cols = kkt.colind.cpu().numpy().astype(np.int64)
        kr, kc = L.blk[rows], L.blk[cols]
        chain_only = (kr >= 0) & (kc >= 0)
        src_c, dest_c = self._chain_plan(rows[chain_only], cols[chain_only])
        src_c = np.nonzero(chain_only)[0][src_c]
        e = (kr >= 0) & (kc < 0)
        tr = kr[e] % Tp
        hc = self.hub_of[cols[e]]
        if (hc // hw != tr).any():
            raise _lib.IemError("chain KKT (hub border): a border unknown couples to a block of another time support")
        lane_e = kr[e] // Tp
        # only a few local rows of a block ever hold border entries: the ones the hubs couple to, and the coupling rows /
        # columns R, C the reduction carries them over — E lives on those rows Q alone (10 of 20 for the SIR model)
        Q = np.union1d(np.union1d(np.unique(L.loc[rows[e]]), L.rowsR), L.colsC).astype(np.int64)
        self.nQ = nQ = int(Q.size)
        qidx = np.full(self.nb, -1, dtype=np.int64); qidx[Q] = np.arange(nQ)
        dest_e = ((tr * self.lanes + lane_e) * nQ + qidx[L.loc[rows[e]]]) * hw + hc % hw       # E0: [Tp, lanes, nQ, hw]
        g = (kr < 0) & (kc < 0)
        dest_g = self.hub_of[rows[g]] * self.Hp + self.hub_of[cols[g]]
        f64 = dict(dtype=t.float64, device=dev)
        as_t = lambda a: t.as_tensor(np.ascontiguousarray(a), device=dev)
        self._src_c, self._dest_c = as_t(src_c), as_t(dest_c)
        self._src_e, self._dest_e = as_t(np.nonzero(e)[0]), as_t(dest_e)
        self._src_g, self._dest_g = as_t(np.nonzero(g)[0]), as_t(dest_g)
        nc = self.nc = int(L.nc)
        self.flat = t.zeros(self.S * self.nb * self.nb + self.S * nc * nc, **f64)
        self.D, self.Bt = self.flat[:self.S * self.nb * self.nb], self.flat[self.S * self.nb * self.nb:]
        self.BR = t.zeros(self.S * nc * nc, **f64)
        self.E0 = t.zeros(Tp * self.lanes * nQ * hw, **f64)
        self._Q = as_t(Q)
        self._qR, self._qC = as_t(qidx[L.rowsR]), as_t(qidx[L.colsC])
        self._q32, self._qR32, self._qC32 = (t.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device=dev) for a in (Q, qidx[L.rowsR], qidx[L.colsC]))
        self.Sbig = t.zeros(self.Hp * self.Hp, **f64)
        pd = L.pad_positions()
        self._pad = as_t(pd[pd < self.S * self.nb * self.nb])          # the unit diagonal of the padding places (chain part)
        virt = np.setdiff1d(np.arange(self.Hp), self.hub_of[border])
        self._virt = as_t(virt * self.Hp + virt)                        # hubs nobody owns (time blocks beyond the last support)
        rt, ct = L.coupling_tables()
        self._rows, self._cols = t.as_tensor(rt, device=dev), t.as_tensor(ct, device=dev)
        self._R = as_t(L.rowsR.astype(np.int64)); self._C = as_t(L.colsC.astype(np.int64))
        self.info = t.zeros(3, dtype=t.int64, device=dev)
        on = np.nonzero(L.blk >= 0)[0]
        self._on, self._pos = as_t(on), as_t(L.blk[on] * self.nb + L.loc[on])
        self._border, self._hub = as_t(border), as_t(self.hub_of[border])
        self._r, self._z = t.zeros(self.S * self.nb, **f64), t.empty(self.S * self.nb, **f64)
        self._levels = levels
        self._chol = None
        self.negative_pivots = None

    def _chain_plan(self, rows, cols):
        """(src, dest) of the chain-to-chain entries into D | Bt (the layout's plan with the border left out)."""
        L = self.layout
        kr, kc, lr, lc = L.blk[rows], L.blk[cols], L.loc[rows], L.loc[cols]
        nb = self.nb
        dest = np.full(rows.size, -1, dtype=np.int64)
        diag, low = kr == kc, kr == kc + 1
        if (np.abs(kr - kc) > 1).any():
            raise _lib.IemError("chain KKT: an entry couples blocks that are not neighbours (the chain grouping does not fit this model)")
        if low.any():
            ri, ci = L._ridx[lr[low]], L._cidx[lc[low]]
            if (ri < 0).any() or (ci < 0).any():
                L.set_coupling(np.union1d(L.rowsR, lr[low]), np.union1d(L.colsC, lc[low]))
                return self._chain_plan(rows, cols)
            dest[low] = self.S * nb * nb + (kr[low] * L.nc + ri) * L.nc + ci
        dest[diag] = (kr[diag] * nb + lr[diag]) * nb + lc[diag]
        src = np.nonzero(dest >= 0)[0]
        return src, dest[src]

    def load(self):
        """Blocks, one-column border and the hubs' own block from the CSR values of the last ``kkt.assemble``."""
        v = self.kkt.vals.to(self.device)
        self.flat.zero_(); self.E0.zero_(); self.Sbig.zero_()
        if self._pad.numel():
            self.flat[self._pad] = 1.0
        self.flat.index_copy_(0, self._dest_c, v[self._src_c])
        self.E0.index_copy_(0, self._dest_e, v[self._src_e])
        self.Sbig.index_copy_(0, self._dest_g, v[self._src_g])
        if self._virt.numel():
            self.Sbig[self._virt] = 1.0
        return self

    # -- one level of the chain's reduction: device kernels, or the caller's restatement ------------------------------------
    def _level(self, s: int, what: int, tiny: float = 1e-30):
        if self._levels is not None:
            return self._levels(self, s, what)
        m = self.model
        p = lambda a: C.c_void_p(a.data_ptr())
        _lib.check(m._L.iem_kkt_chain_level(m._h, self.S, self.Tp, self.nb, self.nc, p(self.D), p(self.Bt), p(self.BR), p(self._rows), p(self._cols),
                                           p(self.info), float(tiny), int(s), int(what)))

    def _hub_level(self, s, E, Z, En, last):
        m = self.model
        p = lambda a: C.c_void_p(a.data_ptr()) if a is not None else None
        _lib.check(m._L.iem_kkt_hub_level(m._h, self.S, self.Tp, self.nb, self.nc, p(self.D), p(self.Bt), p(self._q32), self.nQ, p(self._qR32), int(self._qR32.numel()),
                                         p(self._qC32), int(self._qC32.numel()), self.hw, int(s), p(E), p(Z), p(En), int(last)))

    def factor(self, clip_levels: int = 16, profile: Optional[dict] = None):
        """``profile``: a dict that receives the milliseconds spent per phase (synchronising: diagnosis only)."""
        t = self._torch
        lanes, Tp, nb, nc, hw, Hp, H = self.lanes, self.Tp, self.nb, self.nc, self.hw, self.Hp, self.H
        if self._levels is None:
            self.model._sync_stream()
        import time as _time
        _t0 = [_time.perf_counter()]

        def tick(name):
            if profile is None:
                return
            if self.device != "cpu":
                t.cuda.synchronize()
            now = _time.perf_counter()
            profile[name] = profile.get(name, 0.0) + (now - _t0[0]) * 1e3
            _t0[0] = now
        D4 = self.D.view(lanes, Tp, nb * nb)
        Bt = self.Bt.view(lanes, Tp, nc, nc)
        R, Cc, Q = self._qR, self._qC, self._Q              # positions of the coupling rows / columns inside Q
        qq = (Q[:, None] * nb + Q[None, :]).reshape(-1)
        nR, nC, nQ = int(R.numel()), int(Cc.numel()), self.nQ
        E = self.E0.view(Tp, lanes, nQ, hw).clone()
        self._level(1, 3)
        s = 1
        while s < Tp:
            tick("other")
            self._level(s, 0)                                   # D of the blocks t = (2m+1)s now holds their inverses
            tick("chain eliminate")
            W = (2 * s - 1) * hw
            te = t.arange(s, Tp, 2 * s, device=self.device)      # eliminated: the odd ones of the blocks still alive ...
            ts_ = t.arange(0, Tp, 2 * s, device=self.device)      # ... survivors: the even ones (one more than eliminated when their count is odd)
            n_e, n_s = int(te.numel()), int(ts_.numel())
            Ee = E[1::2]                                        # [n_e, lanes, nQ, W]
            if self._levels is None:                            # Z = D_i^-1 E_i on the rows Q and the survivors' widened columns: two streaming kernels
                Z = t.empty(n_e, lanes, nQ, W, dtype=t.float64, device=self.device)
                En = t.empty(n_s, lanes, nQ, (4 * s - 1) * hw, dtype=t.float64, device=self.device)
                self._hub_level(s, E, Z, En, 0)
            else:                                               # (the same in library calls: what the tests check the kernels' pipeline against)
                Dq = D4[:, s::2 * s][:, :, qq].permute(1, 0, 2).reshape(n_e, lanes, nQ, nQ)
                Z = t.matmul(Dq, Ee)
            A2, Z2 = Ee.reshape(n_e, lanes * nQ, W), Z.reshape(n_e, lanes * nQ, W)
            tick("Z = Dinv E, E widened" if self._levels is None else "Z = Dinv E")
            if n_e > clip_levels:                               # many small intervals: one batched product, written onto the diagonal blocks of S
                Cm = t.bmm(A2.transpose(1, 2), Z2)
                Sv = t.as_strided(self.Sbig, (n_e, W, W), (2 * s * hw * (Hp + 1), Hp, 1), hw * (Hp + 1))
                Sv -= Cm
            else:                                               # few wide ones: clip each to the hubs that exist
                for mi in range(n_e):
                    h0 = (int(te[mi]) - s + 1) * hw
                    w = min(W, H - h0)
                    if w <= 0:
                        continue
                    self._sub_lower(t.as_strided(self.Sbig, (w, w), (Hp, 1), h0 * (Hp + 1)), A2[mi][:, :w], Z2[mi][:, :w], h0)
            tick(f"S accumulate (levels of {'many' if n_e > clip_levels else 'few'} intervals)")
            if self._levels is not None:
                # the survivors t = 2ms: their border columns widen to the hubs of both neighbours
                En = t.zeros(n_s, lanes, nQ, (4 * s - 1) * hw, dtype=t.float64, device=self.device)
                En[..., s * hw: s * hw + W] = E[0::2]
                if n_s > 1:                                         # left neighbour p = j - s (the block eliminated just before j)
                    Bj = Bt[:, 2 * s::2 * s].permute(1, 0, 2, 3)[:, :, :nR, :nC]
                    upd = t.matmul(Bj, Z[:n_s - 1][:, :, Cc, :])    # Bt_j Z_p[C, :]
                    En[1:, :, R, 0:W] = En[1:, :, R, 0:W] - upd
                Bq = Bt[:, s::2 * s].permute(1, 0, 2, 3)[:, :, :nR, :nC]  # K[q, j] on rows R of q = j + s, columns C of j
                updr = t.matmul(Bq.transpose(-1, -2), Z[:, :, R, :])
                En[:n_e, :, Cc, 2 * s * hw: 2 * s * hw + W] = En[:n_e, :, Cc, 2 * s * hw: 2 * s * hw + W] - updr
            E = En
            tick("E update")
            self._level(s, 1)                                   # fold the inverses into the survivors' blocks and couplings
            tick("chain update")
            s *= 2
        # one block per lane is left (t = 0), coupled to every hub (s is the power of two the levels stopped at)
        Ef = E[0][:, :, (s - 1) * hw: (s - 1) * hw + H]
        self._level(1, 2)
        tick("chain lane-final blocks")
        if self._levels is None:
            Zfull = t.empty_like(E)
            self._hub_level(s, E, Zfull, None, 1)
            Zf = Zfull[0][:, :, (s - 1) * hw: (s - 1) * hw + H]
        else:
            Zf = t.matmul(D4[:, 0][:, qq].reshape(lanes, nQ, nQ), Ef)   # [lanes, nQ, H]
        Sv = t.as_strided(self.Sbig, (H, H), (Hp, 1), 0)
        self._sub_lower(Sv, Ef.reshape(lanes * nQ, H), Zf.reshape(lanes * nQ, H), 0)
        tick("S lane-final product")
        self._dense_factor(Sv.contiguous())
        tick("dense block LDL' of the hubs")
        return self

    def _sub_lower(self, Sv, A, Z, h0: int, chunk: int = 960):
        """``Sv -= A' Z`` (symmetric) for the view of ``S`` that starts at hub ``h0``: wide products only into the block lower
        triangle — column chunks cut at the multiples of ``chunk`` hubs, which the pivot blocks of the hubs' LDL' (96) never
        straddle: it reads the lower triangle and the pivot blocks' own squares, nothing else."""
        w = int(Sv.shape[0])
        if w <= 2 * chunk:
            Sv.addmm_(A.transpose(0, 1), Z, alpha=-1.0)
            return
        cuts = [0] + [b - h0 for b in range((h0 // chunk + 1) * chunk, h0 + w, chunk)] + [w]
        for c0, c1 in zip(cuts[:-1], cuts[1:]):
            Sv[c0:, c0:c1].addmm_(A[:, c0:].transpose(0, 1), Z[:, c0:c1], alpha=-1.0)

    # -- the hubs' Schur complement: block LDL' with 96 x 96 pivot blocks inverted by the chain solver's own kernel ------------
    LEAF = 96

    def _dense_factor(self, Sd):
        """``S = L D L'`` in place on the lower triangle, two levels of blocking: pivot blocks of 96 whose inverses come from
        ``kkt_eliminate`` (the Gauss-Jordan inverse on the FP64 matrix cores that inverts the chain's blocks — it also counts
        the pivot signs: the hubs' inertia is MEASURED, not inferred); inside a panel of 480 columns the pivot steps update the
        panel's own columns only, and the rest of the matrix gets ONE update per panel (inner dimension 480, lower triangle in
        column chunks) — library GEMMs.  (rocSOLVER's potrf takes 16.5 ms for 5 000 x 5 000 here — 2 us per column in its
        unblocked panel kernel whatever the blocking, tools/probes/chol_probe.py.)"""
        t = self._torch
        n, NB = int(Sd.shape[0]), self.LEAF
        PW, CH = 5 * NB, 10 * NB
        steps = (n + NB - 1) // NB
        dev = self.device
        self._Lfull = Sd                                            # unit lower block triangle below the pivot blocks, in place
        self._Dinv = t.zeros(steps, NB, NB, dtype=t.float64, device=dev)
        self._dinfo = t.zeros(steps, 3, dtype=t.int64, device=dev)
        eye = t.eye(NB, dtype=t.float64, device=dev)
        # pivot threshold RELATIVE to the largest diagonal entry: a pivot at rounding level of that scale is doubtful, not positive
        dmax = float(Sd.diagonal().abs().max().item())
        tiny = max(1e-30, 1e-14 * dmax) if np.isfinite(dmax) else 1e-30
        Wbuf = t.empty(n, PW, dtype=t.float64, device=dev)          # L21 D of the current panel (= the columns before scaling)
        self._Linv, self._panel = [], PW
        eyeP = t.eye(PW, dtype=t.float64, device=dev)
        m = self.model
        p = lambda a: C.c_void_p(a.data_ptr())
        for p0 in range(0, n, PW):
            p1 = min(p0 + PW, n)
            for k in range(p0, p1, NB):
                e = min(k + NB, p1); w = e - k; ki = k // NB
                blk = self._Dinv[ki]
                blk.copy_(eye)
                blk[:w, :w] = Sd[k:e, k:e]                           # (the last block is padded with a unit diagonal: positive pivots)
                if self._levels is None:
                    _lib.check(m._L.iem_kkt_chain_factor(m._h, 1, NB, -1, 4, p(blk), None, None, None, None, None, None, None, p(self._dinfo[ki]), tiny))   # (ne = -1: the one-block-per-launch shape)
                else:
                    M = blk.clone().numpy()
                    for j in range(NB):
                        self._dinfo[ki, 0] += int(M[j, j] < 0)
                        self._dinfo[ki, 1] += int(not abs(M[j, j]) >= tiny)
                        M[j + 1:, j + 1:] -= np.outer(M[j + 1:, j], M[j, j + 1:]) / M[j, j]
                    blk.copy_(t.linalg.inv(blk))
                if e < n:
                    A21 = Wbuf[e:, k - p0:e - p0]
                    A21.copy_(Sd[e:, k:e])
                    P = A21 @ blk[:w, :w]
                    if e < p1:
                        Sd[e:, e:p1].addmm_(P, A21[:p1 - e].transpose(0, 1), alpha=-1.0)
                    Sd[e:, k:e] = P
            # the panel's own unit lower triangle, inverted once: a solve is then two matrix-vector products per panel instead of a
            # triangular solve over 5 000 dependent rows (1.4 ms each way in the library)
            pw = p1 - p0
            N = Sd[p0:p1, p0:p1] * self._panel_mask(PW)[:pw, :pw]     # L_pp = I + N, N strictly BLOCK lower (its pivot blocks are unit): N^5 = 0
            X = eyeP[:pw, :pw] - N
            for _ in range((pw - 1) // NB - 1):                      # (I + N)^-1 = I - N (I - N (I - N (...)))
                X = eyeP[:pw, :pw] - N @ X
            self._Linv.append(X)
            for c0 in range(p1, n, CH):                              # the rest of the matrix, lower triangle only, once per panel
                c1 = min(c0 + CH, n)
                Sd[c0:, c0:c1].addmm_(Sd[c0:, p0:p1], Wbuf[c0:c1, :p1 - p0].transpose(0, 1), alpha=-1.0)
        self._dense_n, self._dense_steps = n, steps

    def _panel_mask(self, PW):
        """1 below the pivot blocks' own squares of a panel (those hold D_k before its inversion, not L), 0 elsewhere."""
        if getattr(self, "_pmask", None) is None or self._pmask.shape[0] != PW:
            i = np.arange(PW)
            self._pmask = self._torch.as_tensor(((i[:, None] // self.LEAF) > (i[None, :] // self.LEAF)).astype(np.float64), device=self.device)
        return self._pmask

    def _dense_solve(self, b):
        t = self._torch
        n, NB, steps, PW, Sd = self._dense_n, self.LEAF, self._dense_steps, self._panel, self._Lfull
        x = b.clone()
        for pi, p0 in enumerate(range(0, n, PW)):                    # L z = b
            p1 = min(p0 + PW, n)
            xp = self._Linv[pi] @ x[p0:p1]
            x[p0:p1] = xp
            if p1 < n:
                x[p1:].addmv_(Sd[p1:, p0:p1], xp, alpha=-1.0)
        zp = t.zeros(steps * NB, dtype=t.float64, device=self.device); zp[:n] = x
        w = t.bmm(self._Dinv, zp.view(steps, NB, 1)).reshape(-1)[:n].clone()       # D^-1
        for pi, p0 in reversed(list(enumerate(range(0, n, PW)))):    # L' x = w
            p1 = min(p0 + PW, n)
            if p1 < n:
                w[p0:p1].addmv_(Sd[p1:, p0:p1].transpose(0, 1), w[p1:], alpha=-1.0)
            w[p0:p1] = self._Linv[pi].transpose(0, 1) @ w[p0:p1]
        return w

    def inertia(self):
        """``(positive, negative, doubtful)``: the blocks' pivot signs; the hubs' Schur complement counts as all positive when its
        Cholesky factorisation exists, as doubtful otherwise (the caller shifts and factorises again)."""
        info = self.info.cpu().numpy()
        dinfo = self._dinfo.cpu().numpy()
        neg = int(info[0]) + int(dinfo[:, 0].sum())
        n = self.layout.nvar + self.layout.ncon
        return n - neg, neg, int(info[1]) + int(dinfo[:, 1].sum())

    def _chain_solve(self, r):
        """K_c y = r in place (r: S * nb, padded places zero)."""
        if self._levels is not None:
            return self._levels(self, r, "solve")
        m = self.model
        p = lambda a: C.c_void_p(a.data_ptr())
        args = (m._h, self.S, self.Tp, self.nb, 0, self.nc, p(self.D), p(self.Bt), p(self.BR), p(self._rows), p(self._cols), None, p(r), p(self._z), None)
        _lib.check(m._L.iem_kkt_chain_solve_lanes(*args, None, 0))
        _lib.check(m._L.iem_kkt_chain_solve_lanes(*args, None, 1))
        return r

    def solve(self, rhs, profile: Optional[dict] = None):
        t = self._torch
        if self._levels is None:
            self.model._sync_stream()
        import time as _time
        _t0 = [_time.perf_counter()]

        def tick(name):
            if profile is None:
                return
            if self.device != "cpu":
                t.cuda.synchronize()
            now = _time.perf_counter()
            profile[name] = profile.get(name, 0.0) + (now - _t0[0]) * 1e3
            _t0[0] = now
        lanes, Tp, nb, hw, H = self.lanes, self.Tp, self.nb, self.hw, self.H
        Q = self._Q
        E0 = self.E0.view(Tp, lanes, self.nQ, hw)
        r = self._r
        r.zero_(); r[self._pos] = rhs[self._on]
        tick("scatter the right-hand side")
        y = self._chain_solve(r.clone())
        tick("chain solve 1")
        yb = y.view(lanes, Tp, nb)[:, :, Q].permute(1, 0, 2)                          # [Tp, lanes, nQ]
        rB = t.zeros(Tp * hw, dtype=t.float64, device=self.device)
        rB[self._hub] = rhs[self._border]
        rB = rB - (E0 * yb.unsqueeze(-1)).sum((1, 2)).reshape(-1)                     # E' y: hub (t, k) collects its time support's blocks
        tick("E' y")
        xB = self._dense_solve(rB[:H])
        tick("dense solve of the hubs")
        xBp = t.zeros(Tp * hw, dtype=t.float64, device=self.device); xBp[:H] = xB
        corr = (E0 * xBp.view(Tp, 1, 1, hw)).sum(-1)                                  # E x_B: [Tp, lanes, nQ]
        r2 = r.clone()
        r2v = r2.view(lanes, Tp, nb)
        r2v[:, :, Q] = r2v[:, :, Q] - corr.permute(1, 0, 2)
        tick("E x_B")
        x = self._chain_solve(r2)
        tick("chain solve 2")
        out = t.empty_like(rhs)
        out[self._on] = x[self._pos]
        out[self._border] = xBp[self._hub]
        tick("gather the solution")
        return out
