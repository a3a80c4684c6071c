#!/usr/bin/env python3
"""InteriorPointSolver: a few option sets on the examples that do not converge by default.  python tools/probes/ipm_knobs.py"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
warnings.simplefilter("ignore")
from infiniteexamodels.jl_amd import workloads
from infiniteexamodels.jl_amd.backend import ExaTranscriptionBackend
from infiniteexamodels.jl_amd.ipm import InteriorPointSolver
from infiniteexamodels.jl_amd.model import MI355XBackend
knobs = {"default": {}, "delta_c 1e-14": dict(delta_c=1e-14), "delta_c 0": dict(delta_c=0.0), "linear_rtol 1e-12": dict(linear_rtol=1e-12), "refine 2": dict(refine=2), "bound_push 1e-4": dict(bound_push=1e-4, bound_frac=1e-4), "mu_init 1e-3": dict(mu_init=1e-3),
         "bound_push 1e-4, mu_init 1e-4": dict(bound_push=1e-4, bound_frac=1e-4, mu_init=1e-4), "no scaling": dict(nlp_scaling_max_gradient=0.0),
         "merit": dict(line_search="merit"), "max_iter 1000": dict(max_iter=1000), "mu_from_start": dict(mu_from_start=True), "tau 0.9": dict(tau_min=0.9), "kappa_mu 0.5": dict(kappa_mu=0.5, theta_mu=1.2),
         "bound_push 0.1": dict(bound_push=0.1, bound_frac=0.1), "mu_init 10": dict(mu_init=10.0), "mu_init 10, kappa 0.5": dict(mu_init=10.0, kappa_mu=0.5, theta_mu=1.2)}
probs = {"pandemic 20 x 3": lambda be: workloads.pandemic(20, 3, backend=be), "kinetic 20": lambda be: workloads.kinetic_control(20, backend=be),
         "opf 7": lambda be: workloads.opf(7, backend=be), "3-node 50": lambda be: workloads.three_node_design(50, backend=be)}
only = sys.argv[1:]
for pn, build in probs.items():
    if only and pn not in only:
        continue
    for kn, kw in knobs.items():
        try:
            im = build(ExaTranscriptionBackend(InteriorPointSolver(**{**dict(tol=1e-8, max_iter=250), **kw}), backend=MI355XBackend()))
            r = im.optimize()
            print(f"{pn:18s} {kn:24s} {r.status:12s} it {r.iterations:3d}  obj {r.objective:.8g}  err {r.kkt_residual:.1e}", flush=True)
        except Exception as e:       # noqa: BLE001
            print(f"{pn:18s} {kn:24s} FAILED {type(e).__name__}: {str(e)[:100]}", flush=True)
