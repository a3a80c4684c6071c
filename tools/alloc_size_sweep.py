#!/usr/bin/env python3
"""Placement mode vs the SIZE handed to hipMalloc for the two COO value buffers (same process, default code object,
iem_time_kernels x 50): exact, and rounded up to 4 KiB / 64 KiB / 2 MiB (what PyTorch's caching allocator requests) /
2 MiB + 4 KiB / 2 MiB + 1 MiB / 32 MiB / 1 GiB.  Each pair is freed before the next is allocated."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel
import bench
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
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
    assert gm._L.iem_time_kernels(gm._h, xd.data_ptr(), yd.data_ptr(), jp, hp, 50, C.byref(a), C.byref(b)) == 0
    return f"{a.value:.4f}/{b.value:.4f}"


K, M, G = 1 << 10, 1 << 20, 1 << 30
up = lambda n, g: (n + g - 1) // g * g
nj, nh = gm.meta.nnzj * 8, gm.meta.nnzh * 8
for label, f in (("exact", lambda n: n), ("4 KiB", lambda n: up(n, 4 * K)), ("64 KiB", lambda n: up(n, 64 * K)), ("2 MiB", lambda n: up(n, 2 * M)),
                 ("2 MiB + 4 KiB", lambda n: up(n, 2 * M) + 4 * K), ("2 MiB + 1 MiB", lambda n: up(n, 2 * M) + M), ("32 MiB", lambda n: up(n, 32 * M)),
                 ("1 GiB", lambda n: up(n, G)), ("exact again", lambda n: n)):
    j, h = raw(f(nj)), raw(f(nh))
    print(f"{label:14s} jac {f(nj):>11d} B @0x{j:x}  hess {f(nh):>11d} B @0x{h:x}   {timed(j, h)}  {timed(j, h)}", flush=True)
    hip.hipFree(j); hip.hipFree(h)
