#!/usr/bin/env python3
"""Is the small pandemic model (20 + 10 supports x 3 scenarios) a problem SciPy's SQP finds a KKT point of?  (A yardstick for
ipm.InteriorPointSolver, which ends at its iteration limit there.)"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
warnings.simplefilter("ignore")
import numpy as np
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel
from test_gpu_solve import HostView, slsqp
h = HostView(ExaModel(transcribe.exa_core(workloads.pandemic(20, 3)), device=0))
m = h.meta
x0 = np.clip(m.x0, m.lvar + 1e-2, np.where(np.isfinite(m.uvar), m.uvar - 1e-2, np.inf))
t0 = time.time()
r = slsqp(h, x0)
c = h.cons(r.x)
print("SLSQP:", r.status, r.message, "iterations", r.nit, "objective", r.fun, "max violation", max(np.maximum(m.lcon - c, 0).max(), np.maximum(c - m.ucon, 0).max()), f"{time.time() - t0:.1f} s")
