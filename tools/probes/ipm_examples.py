#!/usr/bin/env python3
"""InteriorPointSolver on the device: the reference's test problems (known optima) and examples with bounds / inequality rows.
python tools/probes/ipm_examples.py [name ...]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
warnings.simplefilter("ignore")
import cases
from infiniteexamodels.jl_amd import lib as iemlib, workloads
from infiniteexamodels.jl_amd.backend import ExaTranscriptionBackend
from infiniteexamodels.jl_amd.ipm import InteriorPointSolver
from infiniteexamodels.jl_amd.model import MI355XBackend

MU = bool(int(os.environ.get("IPM_MU_FROM_START", "0")))
mk = lambda **kw: ExaTranscriptionBackend(InteriorPointSolver(tol=1e-8, mu_from_start=MU, line_search=os.environ.get("IPM_LINE_SEARCH", "filter"), **kw), backend=MI355XBackend())
def from_case(fn):
    def build():
        m = fn()
        m = m[0] if isinstance(m, tuple) else m
        m.set_transformation_backend(mk())
        return m
    return build
known = {"rosenbrock": 306.4999755050365, "pfun": 0.48292223509341475, "ode_5x5": -12.784599900757165}
table = {"rosenbrock": from_case(cases.rosenbrock), "pfun": from_case(cases.pfun), "ode_5x5": from_case(cases.ode_5x5),
         "test_problem_1": from_case(cases.test_problem_1), "test_problem_1_oc3": from_case(cases.test_problem_1_oc3),
         "test_problem_2_obj0": from_case(lambda: cases.test_problem_2(0)), "test_problem_2_obj2": from_case(lambda: cases.test_problem_2(2)),
         "farmer 100": lambda: workloads.farmer(100, backend=mk()), "farmer 10000": lambda: workloads.farmer(10000, backend=mk()),
         "opf 7": lambda: workloads.opf(7, backend=mk()), "opf 1000": lambda: workloads.opf(1000, backend=mk()),
         "opf 10000": lambda: workloads.opf(10000, backend=mk()), "farmer 100000": lambda: workloads.farmer(100000, backend=mk()),
         "pandemic 20 x 3": lambda: workloads.pandemic(20, 3, backend=mk()), "pandemic 500 x 3": lambda: workloads.pandemic(500, 3, backend=mk()),
         "kinetic 20": lambda: workloads.kinetic_control(20, backend=mk()), "kinetic 2000": lambda: workloads.kinetic_control(2000, backend=mk()),
         "3-node 50": lambda: workloads.three_node_design(50, backend=mk()), "3-node 5000": lambda: workloads.three_node_design(5000, backend=mk()),
         "quadrotor 2000": lambda: workloads.quadrotor(2000, backend=mk()), "hovercraft 101": lambda: workloads.hovercraft(backend=mk())}
for name in (sys.argv[1:] or list(table)):
    try:
        im = table[name]()
        t0 = time.perf_counter()
        r = im.optimize()
        ms = [h["iteration_ms"] for h in r.history if "iteration_ms" in h]
        print(f"{name:22s} {im.termination_status():18s} it {r.iterations:3d}  obj {r.objective:.10g}  err {r.kkt_residual:.1e}  {time.perf_counter() - t0:.2f} s"
              f"  median iteration {sorted(ms)[len(ms) // 2] if ms else 0:.2f} ms  {known.get(name, '')}", flush=True)
    except Exception as e:       # noqa: BLE001
        print(f"{name:22s} FAILED: {type(e).__name__}: {str(e)[:140]}", flush=True)
