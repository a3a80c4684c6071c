"""Augmented KKT system on the device: assembly from the evaluation path's COO
outputs and factorisation/solve by rocSOLVER's sparse RE-factorisation (the ROCm counterpart of
the CUDSS hand-off the reference uses, ``README.md:36-37``).

The matrix an interior-point solver (MadNLP's sparse augmented system, Ipopt's) factorises at
every iteration is

        K = [ H + Σ + δw·I     Jᵀ   ]        H: Lagrangian Hessian (hess_coord!, lower triangle)
            [      J        −δc·I   ]        J: constraint Jacobian (jac_coord!)

with a sparsity pattern that never changes.  ``KKTSystem``:

* builds the CSR pattern of K ONCE from ``jac_structure!/hess_structure!`` (device tensors, torch
  sort — ``csr.build_plan``) — duplicates of the COO layouts and the two mirrored blocks are summed;
* per iteration, gathers ``[hess | hess (mirrored) | jac | jac (transposed) | Σ+δw | −δc]`` into
  the CSR value array with the one hand-written kernel of the assembly path (``iem_csr_values``);
* ``analyse()``: ONE host factorisation of K at the current point (SuperLU through scipy, COLAMD
  ordering, partial pivoting) gives the fill pattern and the pivot sequence; they are uploaded as
  the bundle matrix T = (L − I) + U with the permutations P, Q (``rocsolver_dcsrrf_analysis``);
* ``factor()`` / ``solve()``: ``rocsolver_dcsrrf_refactlu`` re-factorises the new values on the
  device with the frozen pivot sequence, ``rocsolver_dcsrrf_solve`` does the triangular solves —
  no host round trip.  Frozen pivots are safe here for the same reason they are in MadNLP's GPU
  path: the inertia-correcting regularisation (δw, δc) keeps the quasi-definite K well pivoted.

rocSOLVER/rocBLAS are bound through ctypes (C API, 32-bit indices).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import lib as _lib
from .csr import build_plan

_rs = None


def _rocsolver():
    global _rs
    if _rs is None:
        blas = C.CDLL("librocblas.so", mode=C.RTLD_GLOBAL)
        sol = C.CDLL("librocsolver.so", mode=C.RTLD_GLOBAL)
        _rs = (blas, sol)
    return _rs


def _chk(status: int, what: str):
    if status != 0:
        raise _lib.IemError(f"{what}: rocblas_status {status}")


class KKTSystem:
    def __init__(self, model):
        self.model = m = model
        meta = m.meta
        self.nvar, self.ncon = n, mc = meta.nvar, meta.ncon
        self.n = n + mc
        dev = m.device
        jr, jc = m.jac_structure_device(0)
        hr, hc = m.hess_structure_device(0)
        off = hr != hc                                     # mirrored copy of the strictly lower part
        self._off_idx = torch.nonzero(off).flatten()
        idx = torch.arange
        rows = torch.cat([hr, hc[off], n + jr, jc, idx(n, device=dev), n + idx(mc, device=dev)])
        cols = torch.cat([hc, hr[off], jc, n + jr, idx(n, device=dev), n + idx(mc, device=dev)])
        self.perm, self.seg, rowptr, colind = build_plan(rows, cols, self.n, self.n)
        self.nnz = int(colind.numel())
        if self.nnz >= 2 ** 31 or self.n >= 2 ** 31:
            raise _lib.IemError("KKT system too large for rocSOLVER's 32-bit indices")
        self.rowptr = rowptr.to(torch.int32)
        self.colind = colind.to(torch.int32)
        self.vals = torch.zeros(self.nnz, dtype=torch.float64, device=dev)
        self._n_hess, self._n_jac = int(hr.numel()), int(jr.numel())
        self._coo = torch.empty(int(rows.numel()), dtype=torch.float64, device=dev)
        self._rf = None

    # -- assembly -----------------------------------------------------------------------------
    def assemble(self, hess_vals, jac_vals, sigma=None, delta_w: float = 0.0, delta_c: float = 0.0):
        """CSR values of K from the current hess_coord!/jac_coord! outputs (device tensors)."""
        nh, nj, n, mc = self._n_hess, self._n_jac, self.nvar, self.ncon
        c, o = self._coo, 0
        c[o:o + nh] = hess_vals; o += nh
        k = int(self._off_idx.numel())
        c[o:o + k] = hess_vals[self._off_idx]; o += k
        c[o:o + nj] = jac_vals; o += nj
        c[o:o + nj] = jac_vals; o += nj
        c[o:o + n] = delta_w if sigma is None else sigma + delta_w; o += n
        c[o:o + mc] = -delta_c
        m = self.model
        m._sync_stream()
        _lib.check(m._L.iem_csr_values(m._h, self.nnz, self.seg.data_ptr(), self.perm.data_ptr(), c.data_ptr(),
                                       self.vals.data_ptr()))
        return self.vals

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.vals.cpu().numpy(), self.colind.cpu().numpy(), self.rowptr.cpu().numpy()), shape=(self.n, self.n))

    # -- analysis (host, once per pattern) --------------------------------------------------------
    def analyse(self, nrhs: int = 1):
        """Symbolic analysis + first numeric factorisation on the host at the CURRENT values; fixes
        the fill pattern and the pivot order for every later ``factor()``."""
        import scipy.sparse as sp
        from scipy.sparse.linalg import splu
        blas, sol = _rocsolver()
        K = self.to_scipy().tocsc()
        lu = splu(K, permc_spec="COLAMD", diag_pivot_thresh=0.1)
        n = self.n
        # scipy: Pr·K·Pc = L·U with Pr[perm_r[j], j] = 1, Pc[j, perm_c[j]] = 1
        # rocSOLVER: P·M·Q = L·U with pivP / pivQ = the order in which rows / columns of M were arranged
        pivP = np.argsort(lu.perm_r).astype(np.int32)
        pivQ = np.argsort(lu.perm_c).astype(np.int32)
        T = (lu.L - sp.identity(n, format="csc") + lu.U).tocsr()
        T.sort_indices()
        dev = self.model.device
        self._T_ptr = torch.tensor(T.indptr.astype(np.int32), device=dev)
        self._T_ind = torch.tensor(T.indices.astype(np.int32), device=dev)
        self._T_val = torch.tensor(T.data.astype(np.float64), device=dev)
        self._P = torch.tensor(pivP, device=dev)
        self._Q = torch.tensor(pivQ, device=dev)
        self.nnzT = int(T.nnz)
        self._B = torch.zeros(n * nrhs, dtype=torch.float64, device=dev)
        self._nrhs = nrhs
        if self._rf is None:
            self._handle = C.c_void_p()
            _chk(blas.rocblas_create_handle(C.byref(self._handle)), "rocblas_create_handle")
            self._rf = C.c_void_p()
            _chk(sol.rocsolver_create_rfinfo(C.byref(self._rf), self._handle), "rocsolver_create_rfinfo")
        self._set_stream()
        p = lambda t: C.c_void_p(t.data_ptr())
        _chk(sol.rocsolver_dcsrrf_analysis(self._handle, C.c_int(n), C.c_int(nrhs), C.c_int(self.nnz), p(self.rowptr), p(self.colind),
                                           p(self.vals), C.c_int(self.nnzT), p(self._T_ptr), p(self._T_ind), p(self._T_val),
                                           p(self._P), p(self._Q), p(self._B), C.c_int(n), self._rf), "rocsolver_dcsrrf_analysis")
        return self

    def _set_stream(self):
        blas, _ = _rocsolver()
        s = torch.cuda.current_stream(self.model.device).cuda_stream
        _chk(blas.rocblas_set_stream(self._handle, C.c_void_p(s)), "rocblas_set_stream")

    # -- per iteration (device) -------------------------------------------------------------------
    def factor(self):
        """Numeric re-factorisation of the current ``vals`` on the device (frozen pivot order)."""
        _, sol = _rocsolver()
        self._set_stream()
        p = lambda t: C.c_void_p(t.data_ptr())
        _chk(sol.rocsolver_dcsrrf_refactlu(self._handle, C.c_int(self.n), C.c_int(self.nnz), p(self.rowptr), p(self.colind), p(self.vals),
                                           C.c_int(self.nnzT), p(self._T_ptr), p(self._T_ind), p(self._T_val), p(self._P), p(self._Q),
                                           self._rf), "rocsolver_dcsrrf_refactlu")
        return self

    def solve(self, rhs: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """K·sol = rhs with the current factors; ``rhs`` of length nvar + ncon (device tensor)."""
        _, sol = _rocsolver()
        if rhs.numel() != self.n * self._nrhs:
            raise ValueError("rhs length")
        self._set_stream()
        self._B.copy_(rhs.reshape(-1))
        p = lambda t: C.c_void_p(t.data_ptr())
        _chk(sol.rocsolver_dcsrrf_solve(self._handle, C.c_int(self.n), C.c_int(self._nrhs), C.c_int(self.nnzT), p(self._T_ptr), p(self._T_ind),
                                        p(self._T_val), p(self._P), p(self._Q), p(self._B), C.c_int(self.n), self._rf), "rocsolver_dcsrrf_solve")
        out = out if out is not None else torch.empty_like(self._B)
        out.copy_(self._B)
        return out

    def close(self):
        if self._rf is not None:
            blas, sol = _rocsolver()
            sol.rocsolver_destroy_rfinfo(self._rf)
            blas.rocblas_destroy_handle(self._handle)
            self._rf = None
