"""``ipm.InteriorPointSolver`` — the solver LOGIC on the CPU: the oracle behind the model's method names
(tests/host_model.py) and a dense LDL' factorisation behind the linear-system interface.  The reference's own solver-level
constants are asserted at ITS tolerance (``tol = 1e-6``, /root/reference/test/solve.jl:1) — and Ipopt's iteration count of
the warm-start problem (8, test/ipopt.jl:180) is met, the method being the same one.  The device path (chain KKT solver) is
tests/test_gpu_solve.py."""
import numpy as np
import pytest

import cases
from host_model import HostLinear, HostModel
from infiniteexamodels.jl_amd.backend import ExaTranscriptionBackend
from infiniteexamodels.jl_amd.contrib.ipm import InteriorPointSolver

TOL = 1e-6


class OnTheOracle:
    """the backend's solver slot filled with the interior-point solver, evaluating through the oracle"""
    option_convention = "ipopt"

    def __init__(self, holder, **options):
        self.holder, self.ipm, self.calls = holder, InteriorPointSolver(linear=HostLinear, **options), 0

    def __call__(self, model, x0, y0, **options):
        self.calls += 1
        options.pop("print_level", None)
        return self.ipm(HostModel(self.holder[0].core.to_blob()), x0, y0, **options)


def attach(m, **options):
    holder = []
    be = ExaTranscriptionBackend(None)
    holder.append(be)
    be.set_optimizer(OnTheOracle(holder, **options))
    m.set_transformation_backend(be)
    return be


def test_finite_parameter_problem_and_its_resolve(built):
    """test/solve.jl:134-156"""
    m, (P1, P2) = cases.rosenbrock()
    be = attach(m)
    m.set_silent()
    m.optimize()
    assert m.termination_status() == "LOCALLY_SOLVED" and abs(m.objective_value() - 306.4999755050365) < TOL
    m.set_parameter_value(P1, 90.0)
    m.set_parameter_value(P2, 1.3)
    m.optimize()
    assert abs(m.objective_value() - 276.26497794903645) < TOL and m.value(P1) == 90.0
    x1, x2 = m.infinite_variables[:2]
    assert np.allclose(m.value(x1), 0.5, atol=1e-6) and np.allclose(m.value(x2), 2.0, atol=1e-5)
    # x1 <= 0.5 is active: its dual is non-positive (JuMP's sign), the inactive rows' duals vanish
    c_x1, c_x2 = m.constraints[0], m.constraints[1]
    assert (m.dual(c_x1) < -1e-3).all() and np.allclose(m.dual(c_x2), 0.0, atol=1e-6)


def test_parameter_function_problem(built):
    """test/solve.jl:173-206"""
    m, (pf1, pf2) = cases.pfun()
    attach(m)
    m.optimize()
    assert abs(m.objective_value() - 0.48292223509341475) < TOL
    m.set_parameter_value(pf1, np.cos)
    m.set_parameter_value(pf2, lambda t, s: np.sin(t) * s + 0.8)
    m.optimize()
    assert abs(m.objective_value() - 0.8155916466182952) < TOL
    v = next(q for q in m.infinite_variables if q.name == "v")
    assert (m.dual((v, "lower")) >= -1e-12).all() and (m.dual((v, "upper")) <= 1e-12).all()


def test_warm_start_problem_takes_ipopts_iterations(built):
    """test/ipopt.jl:160-195: -12.784599900757165 in 8 iterations cold; fewer from the previous solution"""
    m = cases.ode_5x5()
    be = attach(m)
    r = m.optimize()
    assert abs(m.objective_value() - (-12.784599900757165)) < TOL and r.iterations == 8
    be.warmstart_backend_start_values()
    r2 = m.optimize()
    assert abs(m.objective_value() - (-12.784599900757165)) < TOL and r2.iterations < r.iterations
    z = m.finite_variables[0]
    y = next(q for q in m.infinite_variables if q.name == "y")
    assert np.isscalar(m.value(z)) and m.value(y).shape == (5, 5) and (m.value(y) >= -1e-7).all()


@pytest.mark.parametrize("name, kw", [("test_problem_1", {}), ("test_problem_1_oc3", {}), ("test_problem_2_obj0", {}), ("test_problem_2_obj1", {}),
                                      ("test_problem_2_obj2", {}), ("pfun_full", {}), ("farmer_5", {"mu_from_start": True}), ("quadrotor_5", {}), ("hovercraft", {})])
def test_kkt_points_of_the_other_small_models(built, name, kw):
    """first-order points (Ipopt's scaled optimality error below 1e-8) re-checked through the oracle: feasibility, bounds,
    stationarity with the returned multipliers, complementarity signs"""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hm = HostModel(cases.build_core(name).to_blob())
    r = InteriorPointSolver(linear=HostLinear, **kw)(hm)
    assert r.status == "first_order", (name, r.status, r.iterations, r.kkt_residual)
    om = hm.om
    x, y, zL, zU = (np.asarray(a) for a in (r.solution, r.multipliers, r.multipliers_L, r.multipliers_U))
    c = om.cons(x)
    scale = 1.0 + np.abs(x).max()
    assert (x >= om.lvar - 1e-7 * scale).all() and (x <= om.uvar + 1e-7 * scale).all()
    assert (c >= om.lcon - 1e-6 * (1 + np.abs(om.lcon))).all() and (c <= om.ucon + 1e-6 * (1 + np.abs(om.ucon))).all()
    big = max(1.0, np.abs(y).max(), np.abs(zL).max(), np.abs(zU).max())
    assert np.abs(om.grad(x) + om.jtprod(x, y) - zL + zU).max() <= 1e-6 * big
    assert (zL >= 0).all() and (zU >= 0).all()


def test_time_limit_and_iteration_limit_statuses(built):
    hm = HostModel(cases.build_core("ode_5x5").to_blob())
    assert InteriorPointSolver(linear=HostLinear, max_iter=2)(hm).status == "max_iter"
    assert InteriorPointSolver(linear=HostLinear, max_wall_time=0.0)(hm).status == "max_time"


@pytest.mark.parametrize("name", ["ode_5x5", "test_problem_2_obj4", "irregular"])
def test_filter_line_search(built, name):
    """the default acceptance rule (Ipopt's filter, the l1 merit function as its fallback): the fifth objective form of
    test/solve.jl:46-90 (non-convex in z) needs it; ``line_search = "merit"`` remains as an option"""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hm = HostModel(cases.build_core(name).to_blob())
    r = InteriorPointSolver(linear=HostLinear)(hm)
    assert r.status == "first_order" and r.iterations <= 40, (r.status, r.iterations, r.kkt_residual)
    if name == "ode_5x5":
        assert InteriorPointSolver(linear=HostLinear, line_search="merit")(hm).status == "first_order"
