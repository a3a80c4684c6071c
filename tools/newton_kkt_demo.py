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
    """Returns (x, y, history); every vector stays on the device.  (The method itself lives in the package:
    ``infiniteexamodels.jl_amd.contrib.newton.LagrangeNewtonSolver`` — what ``ExaTranscriptionBackend`` takes in its solver slot.)"""
    from infiniteexamodels.jl_amd.contrib.newton import LagrangeNewtonSolver
    res = LagrangeNewtonSolver(tol=tol, max_iter=iters, delta_w=delta_w, delta_c=delta_c, log=log)(gm)
    return res.solution, res.multipliers, res.history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--supports", type=int, default=100_000)
    ap.add_argument("--iters", type=int, default=12)
    a = ap.parse_args()
    gm = ExaModel(transcribe.exa_core(workloads.quadrotor(a.supports)), device=0)
    x, y, hist = newton(gm, a.iters, log=lambda h: print(h, flush=True))
    print(json.dumps({"workload": f"quadrotor, {a.supports} supports", "nvar": gm.meta.nvar, "ncon": gm.meta.ncon, "history": hist}))
