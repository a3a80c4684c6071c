#!/usr/bin/env python3
"""What a host that OWNS its COO value buffers could do at set-up (bench.py does not): hipMalloc the two buffers, time the
default kernels into them (iem_time_kernels x 30), and re-draw a buffer (free, malloc again — the driver hands out other
physical pages) while it sits in a slow placement, at most `tries` times.  Prints the draws and the un-synchronised pair
loop (the bench's step) in the first and in the kept buffers."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel
import bench
S, TRIES = 1_000_000, 8
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


def timed(jp, hp, iters=30):
    a, b = C.c_double(), C.c_double()
    assert gm._L.iem_time_kernels(gm._h, xd.data_ptr(), yd.data_ptr(), jp, hp, iters, C.byref(a), C.byref(b)) == 0
    return a.value, b.value


def pair_loop(jp, hp, n=200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        gm._L.iem_jac_coord(gm._h, xd.data_ptr(), jp); gm._L.iem_hess_coord(gm._h, xd.data_ptr(), yd.data_ptr(), 1.0, hp)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        gm._L.iem_jac_coord(gm._h, xd.data_ptr(), jp); gm._L.iem_hess_coord(gm._h, xd.data_ptr(), yd.data_ptr(), 1.0, hp)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


gm._sync_stream()
nj, nh = gm.meta.nnzj * 8, gm.meta.nnzh * 8
jt = torch.empty(gm.meta.nnzj, dtype=torch.float64, device="cuda"); ht = torch.empty(gm.meta.nnzh, dtype=torch.float64, device="cuda")
tj, th = timed(jt.data_ptr(), ht.data_ptr())
print(f"torch.empty buffers: jac {tj:.4f} hess {th:.4f}  pair loop {pair_loop(jt.data_ptr(), ht.data_ptr()):.4f} ms")
best = {}
for which, nbytes, good in (("jac", nj, 0.0850), ("hess", nh, 0.0870)):
    keep, draws = None, []
    for k in range(TRIES):
        p = raw(nbytes)
        other = ht.data_ptr() if which == "jac" else jt.data_ptr()
        t = timed(p, other)[0] if which == "jac" else timed(other, p)[1]
        draws.append(round(t, 4))
        if keep is None or t < keep[1]:
            if keep is not None:
                hip.hipFree(keep[0])
            keep = (p, t)
        else:
            hip.hipFree(p)
        if t <= good:
            break
    best[which] = keep
    print(f"{which}: draws {draws} -> kept {keep[1]:.4f}")
print(f"re-drawn buffers: pair loop {pair_loop(best['jac'][0], best['hess'][0]):.4f} ms = {1e3 / pair_loop(best['jac'][0], best['hess'][0]):.0f} pairs/s")
