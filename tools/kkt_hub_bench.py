#!/usr/bin/env python3
"""Chain KKT solver with a span-sparse HUB border (kkt_chain.HubChainKKT) on BASELINE config 3 — pandemic SIR on a t x xi grid
(ESCAPE34/pandemic.jl), default 5 000 x 100 supports: one chain per scenario, u(t) as hubs.
  python tools/kkt_hub_bench.py [--nt 4990] [--nxi 100] [--iters 5] [--check]"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from infiniteexamodels.jl_amd import lib as iemlib, transcribe, workloads
from infiniteexamodels.jl_amd.kkt import KKTSystem
from infiniteexamodels.jl_amd.kkt_chain import HubChainKKT
from infiniteexamodels.jl_amd.model import ExaModel

ap = argparse.ArgumentParser()
ap.add_argument("--nt", type=int, default=4990)
ap.add_argument("--nxi", type=int, default=100)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--check", action="store_true", help="compare the solution with scipy's sparse LU (small sizes)")
args = ap.parse_args()
t0 = time.perf_counter()
core = transcribe.exa_core(workloads.pandemic(args.nt, args.nxi))
gm = ExaModel(core, device=0)
kkt = KKTSystem(gm)
t1 = time.perf_counter()
hub = HubChainKKT(kkt)
t2 = time.perf_counter()
n = gm.meta.nvar + gm.meta.ncon
rng = np.random.default_rng(0)
x = torch.tensor(np.abs(gm.meta.x0 + 0.1 * rng.standard_normal(gm.meta.nvar)) + 0.05, device="cuda")
y = torch.tensor(0.1 * np.random.default_rng(1).standard_normal(gm.meta.ncon), device="cuda")
sigma = torch.tensor(0.5 + rng.random(gm.meta.nvar), device="cuda")
rhs = torch.tensor(rng.standard_normal(n), device="cuda")
hv, jv = gm.hess_coord(x, y), gm.jac_coord(x)


def timed(fn, iters=args.iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def matvec(v):
    out = torch.empty_like(v)
    gm._sync_stream()
    p = lambda a: C.c_void_p(a.data_ptr())
    iemlib.check(gm._L.iem_csr_spmv(gm._h, kkt.n, p(kkt.rowptr), p(kkt.colind), p(kkt.vals), p(v), p(out), 0, None))
    return out


ms = {"assemble_csr": timed(lambda: kkt.assemble(hv, jv, sigma, 1e-2, 1e-6)), "load_blocks": timed(lambda: hub.load())}
ms["load_and_factor"] = timed(lambda: (hub.load(), hub.factor()))
ms["factor"] = ms["load_and_factor"] - ms["load_blocks"]
hub.load().factor()
ms["solve"] = timed(lambda: hub.solve(rhs))
sol = hub.solve(rhs)
res0 = float((matvec(sol) - rhs).abs().max().item())
sol = sol + hub.solve(rhs - matvec(sol))
res1 = float((matvec(sol) - rhs).abs().max().item())
pos, neg, doubtful = hub.inertia()
prof, sprof = {}, {}
hub.load().factor(profile=prof)
hub.solve(rhs, profile=sprof)
out = {"workload": f"pandemic SIR, {args.nt + 10} x {args.nxi} supports", "n": n, "nnz_K": kkt.nnz,
       "hub_layout": {"lanes": hub.lanes, "blocks_per_lane": hub.Tp, "S": hub.S, "nb": hub.nb, "nc": hub.nc, "hubs": hub.H, "hubs_per_time_block": hub.hw},
       "setup_s": {"model_and_csr_plan": t1 - t0, "hub_layout_and_plan": t2 - t1}, "ms": ms, "factor_phases_ms_synchronised": prof, "solve_phases_ms_synchronised": sprof,
       "abs_residual": {"no_refinement": res0, "one_refinement": res1}, "inertia": [pos, neg, doubtful], "ncon": gm.meta.ncon}
# the same pipeline as ONE object behind the C-ABI (iem_kkt_create in hub mode: analysis and level loop in C++, GEMMs by rocBLAS)
L = gm._L
k = C.c_void_p()
tc = time.perf_counter()
iemlib.check(L.iem_kkt_create(gm._h, 0, C.byref(k)))
create_s = time.perf_counter() - tc
p = lambda a: C.c_void_p(a.data_ptr())
gm._sync_stream()
inertia = (C.c_int64 * 3)()
sol2 = torch.empty_like(rhs)
asm = lambda: iemlib.check(L.iem_kkt_assemble(k, p(hv), p(jv), p(sigma), 1e-2, 1e-6))
cms = {"assemble": timed(asm)}
cms["assemble_factor"] = timed(lambda: (asm(), iemlib.check(L.iem_kkt_factor(k, inertia))))
cms["solve"] = timed(lambda: iemlib.check(L.iem_kkt_solve(k, p(rhs), p(sol2))))
asm(); iemlib.check(L.iem_kkt_factor(k, inertia)); iemlib.check(L.iem_kkt_solve(k, p(rhs), p(sol2)))
cres0 = float((matvec(sol2) - rhs).abs().max().item())
r2 = rhs - matvec(sol2)
d2 = torch.empty_like(rhs)
iemlib.check(L.iem_kkt_solve(k, p(r2), p(d2)))
cres1 = float((matvec(sol2 + d2) - rhs).abs().max().item())
out["c_abi_object"] = {"create_s": create_s, "ms": cms, "assemble_factor_solve_ms": cms["assemble_factor"] + cms["solve"], "inertia": list(inertia),
                       "abs_residual": {"no_refinement": cres0, "one_refinement": cres1}, "max_abs_difference_to_the_python_held_solution": float((sol2 + d2 - sol).abs().max().item())}
iemlib.check(L.iem_kkt_destroy(k))
if args.check:
    import scipy.sparse as sp
    from scipy.sparse.linalg import spsolve
    K = sp.csr_matrix((kkt.vals.cpu().numpy(), kkt.colind.cpu().numpy(), kkt.rowptr.cpu().numpy()), shape=(n, n)).tocsc()
    want = spsolve(K, rhs.cpu().numpy())
    out["rel_error_vs_scipy"] = float(np.abs(sol.cpu().numpy() - want).max() / np.abs(want).max())
print(json.dumps(out))
