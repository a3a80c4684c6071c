#!/usr/bin/env python3
"""Chain KKT solver at scale: assemble -> load dense blocks -> block cyclic reduction -> solve, timed on
the device, residual against the CSR matrix.  `python tools/kkt_chain_bench.py [--workload quadrotor] [--supports 100000]`"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.kkt import KKTSystem
from infiniteexamodels.jl_amd.kkt_chain import ChainKKT
from infiniteexamodels.jl_amd.model import ExaModel

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="quadrotor")
ap.add_argument("--supports", type=int, default=100_000)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--cabi", type=int, default=1, help="also time the iem_kkt_* object")
ap.add_argument("--nxi", type=int, default=8, help="--workload pandemic: scenarios (the time axis has --supports + 10 supports)")
args = ap.parse_args()
mk = {"quadrotor": lambda: workloads.quadrotor(args.supports), "quadrotor_oc3": lambda: workloads.quadrotor(args.supports, collocation=3),
      "farmer": lambda: workloads.farmer(args.supports), "opf": lambda: workloads.opf(args.supports), "hovercraft": lambda: workloads.hovercraft(args.supports),
      "kinetic": lambda: workloads.kinetic_control(args.supports), "pandemic3": lambda: workloads.pandemic(args.supports, 3),
      "pandemic5": lambda: workloads.pandemic(args.supports, 5),
      # a 2-D grid as one chain per scenario, u(t) in the border (kkt_chain: lanes) — the reference's ladder, ESCAPE34/run_cases_gpu.jl:99-102
      "pandemic": lambda: workloads.pandemic(args.supports, args.nxi)}[args.workload]
t0 = time.perf_counter()
core = transcribe.exa_core(mk())
gm = ExaModel(core, device=0)
kkt = KKTSystem(gm)
t1 = time.perf_counter()
ck = ChainKKT(kkt)
t2 = time.perf_counter()
L = ck.layout
n = gm.meta.nvar + gm.meta.ncon
rng = np.random.default_rng(0)
x = torch.tensor(gm.meta.x0 + 0.1 * rng.standard_normal(gm.meta.nvar) if args.workload not in ("farmer", "pandemic3", "pandemic5", "pandemic") else np.abs(gm.meta.x0 + 0.1 * rng.standard_normal(gm.meta.nvar)) + 0.05, device="cuda")
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


ms = {"assemble_csr": timed(lambda: kkt.assemble(hv, jv, sigma, 1e-2, 1e-6)), "load_blocks": timed(lambda: ck.load())}
ms["factor"] = timed(lambda: (ck.load(), ck.factor()))- ms["load_blocks"]
ck.load().factor()
ms["solve_1_refinement"] = timed(lambda: ck.solve(rhs, refine=1))
ms["solve_no_refinement"] = timed(lambda: ck.solve(rhs, refine=0))
sol = ck.solve(rhs, refine=1)
res = float((ck._matvec(sol) - rhs).abs().max().item() / max(1.0, float(rhs.abs().max().item())))
res0 = float((ck._matvec(ck.solve(rhs, refine=0)) - rhs).abs().max().item())
pos, neg, doubtful = ck.inertia()
# the same solver as ONE C-ABI object (iem_kkt_*): analysis in C++, blocks filled straight from the COO value buffers
import ctypes as C
from infiniteexamodels.jl_amd import lib as iemlib
cabi = {}
if args.cabi:
    k = C.c_void_p()
    t3 = time.perf_counter()
    iemlib.check(gm._L.iem_kkt_create(gm._h, 0, C.byref(k)))
    cabi["create_s"] = time.perf_counter() - t3
    p = lambda t: C.c_void_p(t.data_ptr())
    inertia = (C.c_int64 * 3)()
    out = torch.empty_like(rhs)
    gm._sync_stream()
    asm = lambda: iemlib.check(gm._L.iem_kkt_assemble(k, p(hv), p(jv), p(sigma), 1e-2, 1e-6))
    cabi["assemble_ms"] = timed(asm)
    cabi["assemble_factor_ms"] = timed(lambda: (asm(), iemlib.check(gm._L.iem_kkt_factor(k, inertia))))
    cabi["solve_ms"] = timed(lambda: iemlib.check(gm._L.iem_kkt_solve(k, p(rhs), p(out))))
    torch.cuda.synchronize()
    cabi["inertia"] = [int(v) for v in inertia]
    cabi["abs_residual_without_refinement"] = float((ck._matvec(out) - rhs).abs().max().item())
    iemlib.check(gm._L.iem_kkt_destroy(k))
flops = L.S * (2.0 * L.nb ** 3 * (1 + 2 + 3) + 2.0 * L.nb * L.nb * L.ne * 4)      # inverse + X, Y + three update products (+ border terms)
print(json.dumps({"workload": args.workload, "supports": args.supports, "n": n, "nnz_K": kkt.nnz, "chain": {"S": L.S, "nb": L.nb, "ne": L.ne, "nc": L.nc, "reach": L.reach, "group": L.group, "phase": L.phase, "lanes": getattr(L, "lanes", 1)},
                  "setup_s": {"model_and_csr_plan": t1 - t0, "chain_layout_and_plan": t2 - t1}, "ms": ms,
                  "factor_GFLOP_dense": flops / 1e9, "factor_TFLOPs": flops / (ms["factor"] * 1e-3) / 1e12,
                  "block_bytes": int(ck.flat.numel() * 8 + (ck.BR.numel() + ck.Z.numel()) * 8),
                  "rel_residual_after_refinement": res, "abs_residual_without": res0, "inertia": [pos, neg, doubtful], "ncon": gm.meta.ncon, "c_abi_object": cabi}))
