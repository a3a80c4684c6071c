#!/usr/bin/env python3
"""One interior-point iteration's worth of device work after the evaluation path:
hess_coord!+jac_coord! -> KKT assembly (iem_csr_values) -> rocSOLVER re-factorisation -> solve.

  python tools/kkt_bench.py [--workload quadrotor] [--supports 20000] [--iters 20]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.kkt import KKTSystem
from infiniteexamodels.jl_amd.model import ExaModel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="quadrotor")
    ap.add_argument("--supports", type=int, default=20000)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    im = {"quadrotor": lambda: workloads.quadrotor(a.supports), "opf": lambda: workloads.opf(a.supports),
          "quadrotor_oc3": lambda: workloads.quadrotor(a.supports, collocation=3)}[a.workload]()
    gm = ExaModel(transcribe.exa_core(im), device=0)
    n, m = gm.meta.nvar, gm.meta.ncon
    rng = np.random.default_rng(0)
    x = torch.tensor(gm.meta.x0 + 0.1 * rng.standard_normal(n), device="cuda")
    y = torch.tensor(rng.standard_normal(m), device="cuda")
    sigma = torch.tensor(0.5 + rng.random(n), device="cuda")
    rhs = torch.tensor(rng.standard_normal(n + m), device="cuda")
    t0 = time.perf_counter(); kkt = KKTSystem(gm); t_plan = time.perf_counter() - t0
    hv = torch.empty(gm.meta.nnzh, dtype=torch.float64, device="cuda")
    jv = torch.empty(gm.meta.nnzj, dtype=torch.float64, device="cuda")
    gm.hess_coord(x, y, hv, obj_weight=1.0); gm.jac_coord(x, jv)
    kkt.assemble(hv, jv, sigma, 1e-2, 1e-6)
    t0 = time.perf_counter(); kkt.analyse(); torch.cuda.synchronize(); t_analyse = time.perf_counter() - t0

    def timed(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters
    ms = {"eval (hess_coord!+jac_coord!)": timed(lambda: (gm.hess_coord(x, y, hv, obj_weight=1.0), gm.jac_coord(x, jv))),
          "assemble K (csr gather-sum)": timed(lambda: kkt.assemble(hv, jv, sigma, 1e-2, 1e-6)),
          "refactor (rocsolver csrrf_refactlu)": timed(kkt.factor),
          "solve (rocsolver csrrf_solve)": timed(lambda: kkt.solve(rhs))}
    sol = kkt.solve(rhs)
    K = kkt.to_scipy()
    res = float(np.abs(K @ sol.cpu().numpy() - rhs.cpu().numpy()).max())
    print(json.dumps({"workload": f"{a.workload}, {a.supports} supports", "n": n + m, "nnz_K": kkt.nnz, "nnz_LU": kkt.nnzT,
                      "plan_s": t_plan, "host_analysis_s": t_analyse, "ms": ms, "residual_inf": res}))


if __name__ == "__main__":
    main()
