"""COO → CSR assembly on the device.

``jac_coord!`` / ``hess_coord!`` produce ExaModels' COO layout, which may repeat positions;
sparse direct solvers (the reference pairs the GPU path with CUDSS, ``README.md:36-37``) want
CSR/CSC with duplicates summed.  The plan — sort of the COO positions by (row, col), segment
boundaries, ``rowptr`` / ``colind`` — is built ONCE per model from the device-generated
structure with torch primitives (sort, unique_consecutive); the per-iteration part is one
hand-written HIP kernel (``iem_csr_gather_sum``: one thread per CSR nonzero, fixed summation
order, no atomics) reached through ``iem_csr_values``.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as _lib


def build_plan(rows: torch.Tensor, cols: torch.Tensor, nrows: int, ncols: int):
    """rows/cols: 0-based int64 tensors (any device).  Returns (perm, seg, rowptr, colind)."""
    key = rows * ncols + cols
    skey, perm = torch.sort(key, stable=True)
    ukey, counts = torch.unique_consecutive(skey, return_counts=True)
    seg = torch.zeros(ukey.numel() + 1, dtype=torch.int64, device=key.device)
    torch.cumsum(counts, 0, out=seg[1:])
    urow = torch.div(ukey, ncols, rounding_mode="floor")
    colind = ukey - urow * ncols
    rowptr = torch.zeros(nrows + 1, dtype=torch.int64, device=key.device)
    torch.cumsum(torch.bincount(urow, minlength=nrows), 0, out=rowptr[1:])
    return perm.contiguous(), seg, rowptr, colind.contiguous()


class CsrAssembler:
    """``which`` = "jac" (ncon × nvar) or "hess" (nvar × nvar, lower triangle as produced)."""

    def __init__(self, model, which: str = "jac"):
        self.model = model
        if which == "jac":
            r, c = model.jac_structure_device(0)
            self.shape = (model.meta.ncon, model.meta.nvar)
        elif which == "hess":
            r, c = model.hess_structure_device(0)
            self.shape = (model.meta.nvar, model.meta.nvar)
        else:
            raise ValueError(which)
        self.perm, self.seg, self.rowptr, self.colind = build_plan(r, c, *self.shape)
        self.nnz = int(self.colind.numel())
        self.n_coo = int(r.numel())
        # 32-bit plan words when every COO position fits: the plan is half of what the assembly kernel reads
        # (torch has no uint32 arithmetic; int32 holds the same bits for values < 2^31, which is the bound used)
        self.narrow = self.n_coo < 2 ** 31
        if self.narrow:
            self.perm32, self.seg32 = self.perm.to(torch.int32), self.seg.to(torch.int32)

    def values(self, coo_vals: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """CSR values of the current COO values (duplicates summed)."""
        m = self.model
        if coo_vals.numel() != self.n_coo or coo_vals.dtype != torch.float64 or not coo_vals.is_cuda:
            raise ValueError("coo_vals must be the float64 CUDA output of jac_coord/hess_coord")
        out = out if out is not None else torch.empty(self.nnz, dtype=torch.float64, device=coo_vals.device)
        m._sync_stream()
        if self.narrow:
            _lib.check(m._L.iem_csr_values32(m._h, self.nnz, self.seg32.data_ptr(), self.perm32.data_ptr(),
                                             coo_vals.data_ptr(), out.data_ptr()))
        else:
            _lib.check(m._L.iem_csr_values(m._h, self.nnz, self.seg.data_ptr(), self.perm.data_ptr(),
                                           coo_vals.data_ptr(), out.data_ptr()))
        return out

    def torch_csr(self, coo_vals: torch.Tensor):
        return torch.sparse_csr_tensor(self.rowptr, self.colind, self.values(coo_vals), size=self.shape)
