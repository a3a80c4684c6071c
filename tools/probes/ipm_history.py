import os, sys, warnings
sys.path[:0] = ['.', 'tests']
warnings.simplefilter("ignore")
from infiniteexamodels.jl_amd import workloads
from infiniteexamodels.jl_amd.backend import ExaTranscriptionBackend
from infiniteexamodels.jl_amd.ipm import InteriorPointSolver
from infiniteexamodels.jl_amd.model import MI355XBackend
name, iters = sys.argv[1], int(sys.argv[2])
mk = lambda: ExaTranscriptionBackend(InteriorPointSolver(tol=1e-8, max_iter=iters), backend=MI355XBackend())
im = {"pandemic": lambda: workloads.pandemic(20, 3, backend=mk()), "farmer": lambda: workloads.farmer(100, backend=mk()),
      "kinetic": lambda: workloads.kinetic_control(20, backend=mk()), "opf": lambda: workloads.opf(7, backend=mk())}[name]()
r = im.optimize()
hs = r.history
for h in hs:
    print(" ".join(f"{k}={(f'{v:.3g}' if isinstance(v, float) else v)}" for k, v in h.items() if k in ("iter", "obj", "dual_inf", "primal_inf", "compl", "mu", "step", "step_dual", "delta_w", "factorisations")))
