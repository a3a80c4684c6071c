#!/usr/bin/env python3
"""Chain KKT kernels on SYNTHETIC well-conditioned blocks, for a range of block sizes (every tile / wave layout of
csrc/iem_kkt_device.h): inverse, products and updates against the numpy restatement (tests/chain_reference.py).  A logic
error shows as an O(1) difference, rounding as 1e-13."""
import ctypes as C_, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import chain_reference as ref
from infiniteexamodels.jl_amd import transcribe, workloads, lib as _lib
from infiniteexamodels.jl_amd.model import ExaModel

gm = ExaModel(transcribe.exa_core(workloads.quadrotor(10)), device=0)
out = []
import chain_reference as ref
for nb, ne, nc in [(12, 0, 4), (16, 0, 8), (20, 0, 8), (40, 0, 12), (48, 0, 16), (52, 0, 12), (64, 0, 24), (68, 0, 20), (84, 0, 20), (96, 0, 48), (12, 4, 4), (40, 8, 12), (60, 52, 8), (84, 16, 32)]:
    rng = np.random.default_rng(nb * 100 + ne)
    S, nv = 11, (nb * 5) // 9
    # coupling rows R among the constraint rows, columns C among the variables (nR, nC < nc: the padding is exercised)
    nR, nC = nc - 1, max(nc - 2, 1)
    R = np.sort(rng.choice(np.arange(nv, nb), size=min(nR, nb - nv), replace=False)); C = np.sort(rng.choice(np.arange(nv), size=min(nC, nv), replace=False))
    D = np.zeros((S, nb, nb)); B = np.zeros((S, nb, nb)); E = 0.3 * rng.standard_normal((S, nb, ne)) / np.sqrt(nb)
    Bt = np.zeros((S, nc, nc))
    for k in range(S):
        H = rng.standard_normal((nv, nv)); H = H @ H.T / nv + 2.0 * np.eye(nv)
        J = rng.standard_normal((nb - nv, nv)) / np.sqrt(nv)
        D[k, :nv, :nv] = H; D[k, nv:, :nv] = J; D[k, :nv, nv:] = J.T; D[k, nv:, nv:] = -0.5 * np.eye(nb - nv)
        if k:
            Bt[k, :R.size, :C.size] = 0.5 * rng.standard_normal((R.size, C.size))
            B[k][R[:, None], C[None, :]] = Bt[k, :R.size, :C.size]
    Dinv, X, Y, Z, Gp, neg_ref = ref.factor(D, B, E)
    rhs = rng.standard_normal((S, nb)); rB = rng.standard_normal(ne)
    G = 3.0 * np.eye(ne) if ne else np.zeros((0, 0))
    xs_ref, xB_ref = ref.solve(Dinv, X, Y, Z, G, Gp, rhs, rB)
    t = lambda a, dt=None: torch.tensor(np.ascontiguousarray(a), device="cuda", dtype=dt)
    rt = np.full(nc, -1, np.int32); rt[:R.size] = R; ct = np.full(nc, -1, np.int32); ct[:C.size] = C
    dD, dBt, dE, dBR, drows, dcols = t(D), t(Bt), t(E if ne else np.zeros(1)), t(np.zeros((S, nc, nc))), t(rt), t(ct)
    dZ = t(np.zeros((S, nb, max(ne, 1)))); dGp = t(np.zeros((S, max(ne, 1), max(ne, 1)))); info = torch.zeros(4, dtype=torch.int64, device="cuda")
    p = lambda a: C_.c_void_p(a.data_ptr())
    gm._sync_stream()
    _lib.check(gm._L.iem_kkt_chain_factor(gm._h, S, nb, ne, nc, p(dD), p(dBt), p(dBR), p(drows), p(dcols), p(dE), p(dZ), p(dGp), p(info), 1e-30))
    torch.cuda.synchronize()
    rel = lambda a, b: float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
    r = dict(nb=nb, ne=ne, nc=nc, Dinv=rel(dD.cpu().numpy(), Dinv), neg=(int(info[0]), int(neg_ref)))
    if ne:
        r["Z"] = rel(dZ.cpu().numpy()[:, :, :ne], Z); r["Gp"] = rel(dGp.cpu().numpy()[:, :ne, :ne], Gp)
    # solve
    dr, dz, drBp = t(rhs), t(np.zeros((S, nb))), t(np.zeros((S, max(ne, 1))))
    args = (gm._h, S, nb, ne, nc, p(dD), p(dBt), p(dBR), p(drows), p(dcols), p(dZ), p(dr), p(dz), p(drBp))
    _lib.check(gm._L.iem_kkt_chain_solve(*args, None, 0))
    xB = None
    if ne:
        Gs = G - dGp.cpu().numpy()[:, :ne, :ne].sum(0)
        xBh = np.linalg.solve(Gs, rB - drBp.cpu().numpy()[:, :ne].sum(0))
        xB = t(xBh); r["xB"] = rel(xBh, xB_ref)
    _lib.check(gm._L.iem_kkt_chain_solve(*args, p(xB) if xB is not None else None, 1))
    torch.cuda.synchronize()
    r["x"] = rel(dr.cpu().numpy(), xs_ref)
    out.append(r); print(r, flush=True)
print(json.dumps(out))
