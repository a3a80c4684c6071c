#!/usr/bin/env python3
"""The iem_kkt_* object in hub mode at BASELINE config 3 (pandemic 5 000 x 100), nothing else: for rocprofv3 --kernel-trace --stats
(which launches make up iem_kkt_factor / iem_kkt_solve).  python tools/kkt_cabi_hub_profile.py [--nt 4990] [--nxi 100] [--iters 5]"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from infiniteexamodels.jl_amd import lib as iemlib, transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel

ap = argparse.ArgumentParser()
ap.add_argument("--nt", type=int, default=4990)
ap.add_argument("--nxi", type=int, default=100)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--what", default="factor,solve")
args = ap.parse_args()
gm = ExaModel(transcribe.exa_core(workloads.pandemic(args.nt, args.nxi)), device=0)
n = gm.meta.nvar + gm.meta.ncon
rng = np.random.default_rng(0)
x = torch.tensor(np.abs(gm.meta.x0 + 0.1 * rng.standard_normal(gm.meta.nvar)) + 0.05, device="cuda")
y = torch.tensor(0.1 * np.random.default_rng(1).standard_normal(gm.meta.ncon), device="cuda")
sigma = torch.tensor(0.5 + rng.random(gm.meta.nvar), device="cuda")
rhs = torch.tensor(rng.standard_normal(n), device="cuda")
hv, jv = gm.hess_coord(x, y), gm.jac_coord(x)
L, k = gm._L, C.c_void_p()
iemlib.check(L.iem_kkt_create(gm._h, 0, C.byref(k)))
p = lambda a: C.c_void_p(a.data_ptr())
gm._sync_stream()
inertia = (C.c_int64 * 3)()
sol = torch.empty_like(rhs)


def timed(fn, iters=args.iters):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


out = {}
asm = lambda: iemlib.check(L.iem_kkt_assemble(k, p(hv), p(jv), p(sigma), 1e-2, 1e-6))
asm(); iemlib.check(L.iem_kkt_factor(k, inertia))
if "factor" in args.what:
    out["assemble_ms"] = timed(asm)
    out["assemble_factor_ms"] = timed(lambda: (asm(), iemlib.check(L.iem_kkt_factor(k, inertia))))
if "solve" in args.what:
    out["solve_ms"] = timed(lambda: iemlib.check(L.iem_kkt_solve(k, p(rhs), p(sol))))
out["inertia"] = list(inertia)
print(json.dumps(out))
