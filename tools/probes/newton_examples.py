#!/usr/bin/env python3
"""LagrangeNewtonSolver on the reference's examples that are equality-constrained without bounds (the others are refused):
which converge from the examples' own start values, in how many iterations.  python tools/probes/newton_examples.py"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.simplefilter("ignore")
from infiniteexamodels.jl_amd import lib as iemlib, workloads
from infiniteexamodels.jl_amd.backend import ExaTranscriptionBackend
from infiniteexamodels.jl_amd.model import MI355XBackend
from infiniteexamodels.jl_amd.contrib.newton import LagrangeNewtonSolver

mk = lambda: ExaTranscriptionBackend(LagrangeNewtonSolver(tol=1e-8, max_iter=60), backend=MI355XBackend())
cases = {"quadrotor 2000": lambda: workloads.quadrotor(2000, backend=mk()),
         "quadrotor 2000, collocation 3": lambda: workloads.quadrotor(2000, collocation=3, backend=mk()),
         "hovercraft 101": lambda: workloads.hovercraft(backend=mk()),
         "hovercraft 2001": lambda: workloads.hovercraft(2001, backend=mk()),
         "hovercraft 21, collocation 4": lambda: workloads.hovercraft(21, collocation=4, backend=mk()),
         "kinetic control 200": lambda: workloads.kinetic_control(200, backend=mk()),
         "3-node design 50": lambda: workloads.three_node_design(50, backend=mk()),
         "farmer 100": lambda: workloads.farmer(100, backend=mk()),
         "pandemic 200 x 3": lambda: workloads.pandemic(200, 3, backend=mk())}
for name, build in cases.items():
    try:
        im = build()
        t0 = time.perf_counter()
        r = im.optimize()
        print(f"{name:32s} {im.termination_status():16s} iterations {r.iterations:3d}  kkt {r.kkt_residual:.2e}  objective {r.objective:.9g}  {time.perf_counter() - t0:.2f} s", flush=True)
    except (iemlib.IemError, TypeError) as e:
        print(f"{name:32s} refused: {str(e)[:110]}", flush=True)
