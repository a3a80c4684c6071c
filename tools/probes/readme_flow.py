import sys, time; sys.path.insert(0, '.')
from infiniteexamodels.jl_amd import workloads
from infiniteexamodels.jl_amd.backend import ExaTranscriptionBackend
from infiniteexamodels.jl_amd.model import MI355XBackend
from infiniteexamodels.jl_amd.contrib.newton import LagrangeNewtonSolver
t0 = time.time()
im = workloads.quadrotor(100_000, backend=ExaTranscriptionBackend(LagrangeNewtonSolver(tol=1e-8), backend=MI355XBackend()))
im.set_silent()
im.optimize()
print(im.termination_status(), im.objective_value(), im.value(im.infinite_variables[0]).shape, im.dual(im.constraints[0]).shape, "solve_time", im.solve_time(), "total", time.time() - t0)
# a re-solve (start values moved, as an MPC loop would): the KKT set-up is kept on the model
im.set_start_value(im.infinite_variables[0], 0.1)
im.optimize()
print("re-solve:", im.termination_status(), "solve_time", im.solve_time(), "iterations", im.backend.results.iterations)
