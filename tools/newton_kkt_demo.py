#!/usr/bin/env python3
"""A solver iteration end to end ON THE DEVICE (SURVEY §8 f3 + f4): the five evaluation calls, the KKT assembly and the
chain factorisation / solve — no host round trip except the scalars a line search looks at.  Equality-constrained Newton
(Lagrange-Newton / SQP with exact Hessian, regularised like an interior-point augmented system, backtracking on the KKT
residual) on the quadrotor tracking problem (examples/quadrotor.jl): what MadNLP does per iteration at the reference's
plug point (ext/InfiniteExaModelsMadNLP.jl:49-50,64), minus its barrier and filter logic.  NOT a product solver — a
demonstration that the hand-off works and what one iteration costs.

  python tools/newton_kkt_demo.py [--supports 100000] [--iters 12]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.kkt import KKTSystem
from infiniteexamodels.jl_amd.kkt_chain import ChainKKT
from infiniteexamodels.jl_amd.model import ExaModel


def newton(gm, iters=12, delta_w=1e-8, delta_c=1e-10, tol=1e-8, log=None):
    """Returns (x, y, history); every vector stays on the device."""
    n, m = gm.meta.nvar, gm.meta.ncon
    kkt = KKTSystem(gm)
    ck = ChainKKT(kkt)
    dev = gm.device
    x = torch.tensor(gm.meta.x0, device=dev)
    y = torch.zeros(m, dtype=torch.float64, device=dev)
    lcon = torch.tensor(gm.meta.lcon, device=dev)      # equality rows: lcon == ucon
    g, c = torch.empty(n, dtype=torch.float64, device=dev), torch.empty(m, dtype=torch.float64, device=dev)
    jv, hv = torch.empty(gm.meta.nnzj, dtype=torch.float64, device=dev), torch.empty(gm.meta.nnzh, dtype=torch.float64, device=dev)
    jtv = torch.empty(n, dtype=torch.float64, device=dev)
    hist = []

    def residual(x, y):
        gm.grad(x, g); gm.cons(x, c); gm.jtprod(x, y, jtv)
        return torch.cat([g + jtv, c - lcon])

    r = residual(x, y)
    for it in range(iters):
        rn = float(r.abs().max().item())
        hist.append(dict(iter=it, kkt_residual=rn, obj=gm.obj(x)))
        if log:
            log(hist[-1])
        if rn <= tol:
            break
        t0 = time.perf_counter()
        gm.jac_hess_coord(x, y, jv, hv, obj_weight=1.0)
        # inertia correction as in Ipopt / MadNLP: the factorisation reports the pivot signs; while they are not
        # (nvar, ncon, 0) the Hessian block is shifted by a growing delta_w and the system factorised again
        dw, tries = delta_w, 0
        while True:
            kkt.assemble(hv, jv, None, dw, delta_c)
            ck.load().factor()
            pos, neg, doubtful = ck.inertia()
            tries += 1
            if (neg == m and doubtful == 0) or tries >= 12:
                break
            dw = max(1e-4, dw * 10.0)
        d = ck.solve(-r, refine=1)
        dx, dy = d[:n], d[n:]
        step = 1.0
        for _ in range(20):                               # backtracking on the KKT residual
            rt = residual(x + step * dx, y + step * dy)
            if float(rt.abs().max().item()) < rn or step < 1e-6:
                break
            step *= 0.5
        x, y, r = x + step * dx, y + step * dy, rt
        torch.cuda.synchronize()
        hist[-1].update(step=step, iteration_ms=(time.perf_counter() - t0) * 1e3, inertia=(pos, neg, doubtful), delta_w=dw, factorisations=tries)
    kkt.close()
    return x, y, hist


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--supports", type=int, default=100_000)
    ap.add_argument("--iters", type=int, default=12)
    a = ap.parse_args()
    gm = ExaModel(transcribe.exa_core(workloads.quadrotor(a.supports)), device=0)
    x, y, hist = newton(gm, a.iters, log=lambda h: print(h, flush=True))
    print(json.dumps({"workload": f"quadrotor, {a.supports} supports", "nvar": gm.meta.nvar, "ncon": gm.meta.ncon, "history": hist}))
