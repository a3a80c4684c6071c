#!/usr/bin/env python3
"""Does the placement mode depend on HOW the caller allocates its COO buffers?  One bench-like process; the default code
object timed (iem_time_kernels x 50, two rounds) into jac/hess buffers obtained by: torch.empty (caching allocator),
plain hipMalloc, hipMalloc of 1-GiB-rounded sizes, and hipMalloc in the opposite order (hess first)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel
import bench
S = 1_000_000
core = transcribe.exa_core(workloads.quadrotor(S))
gm = ExaModel(core, device=0)
x, y = bench.eval_point(gm.meta.nvar, gm.meta.ncon, gm.meta.x0, S, seed=0)
xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]


def raw(nbytes):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), nbytes) == 0
    return p.value


def timed(jp, hp):
    a, b = C.c_double(), C.c_double()
    out = []
    for _ in range(2):
        assert gm._L.iem_time_kernels(gm._h, xd.data_ptr(), yd.data_ptr(), jp, hp, 50, C.byref(a), C.byref(b)) == 0
        out.append(f"{a.value:.4f}/{b.value:.4f}")
    return " ".join(out)


nj, nh = gm.meta.nnzj * 8, gm.meta.nnzh * 8
jt = torch.empty(gm.meta.nnzj, dtype=torch.float64, device="cuda"); ht = torch.empty(gm.meta.nnzh, dtype=torch.float64, device="cuda")
print("torch.empty          ", timed(jt.data_ptr(), ht.data_ptr()))
j1, h1 = raw(nj), raw(nh)
print("hipMalloc jac,hess   ", timed(j1, h1))
G = 1 << 30
j2, h2 = raw((nj + G - 1) // G * G), raw((nh + G - 1) // G * G)
print("hipMalloc 1GiB-sized ", timed(j2, h2))
h3, j3 = raw(nh), raw(nj)
print("hipMalloc hess,jac   ", timed(j3, h3))
print("torch.empty again    ", timed(jt.data_ptr(), ht.data_ptr()))
