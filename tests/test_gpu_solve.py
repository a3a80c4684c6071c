"""Solver-in-the-loop on the GPU path: scipy drives the NLPModels surface of the DEVICE model
(every obj / grad! / cons! / jac_coord! / hess_coord! call goes through the C-ABI to the HIP
kernels) and must reach the reference's known optima (test/solve.jl:146,154,187;
test/ipopt.jl:181) — the same end-to-end pin the reference's own tests use."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import Bounds, NonlinearConstraint, minimize

import cases
from infiniteexamodels.jl_amd import transcribe

pytestmark = pytest.mark.gpu


class HostView:
    """numpy façade over a device ExaModel (host<->device copies per call; tiny problems)."""

    def __init__(self, gm):
        import torch
        self.gm, self.torch = gm, torch
        self.meta = gm.meta

    def _d(self, a):
        return self.torch.tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")

    def obj(self, x):
        return self.gm.obj(self._d(x))

    def grad(self, x):
        return self.gm.grad(self._d(x)).cpu().numpy()

    def cons(self, x):
        return self.gm.cons(self._d(x)).cpu().numpy()

    def jac(self, x):
        r, c = self.gm.jac_structure()
        return sp.coo_matrix((self.gm.jac_coord(self._d(x)).cpu().numpy(), (r, c)),
                             shape=(self.meta.ncon, self.meta.nvar)).tocsr()

    def hess(self, x, y, w):
        r, c = self.gm.hess_structure()
        v = self.gm.hess_coord(self._d(x), self._d(y), obj_weight=w).cpu().numpy()
        L = sp.coo_matrix((v, (r, c)), shape=(self.meta.nvar, self.meta.nvar)).tocsr()
        return (L + L.T - sp.diags(L.diagonal())).tocsr()


def slsqp(h: HostView, x0):
    m = h.meta
    lc, uc = m.lcon, m.ucon
    eq = np.nonzero(lc == uc)[0]
    lo = np.nonzero((lc > -np.inf) & (lc != uc))[0]
    up = np.nonzero((uc < np.inf) & (lc != uc))[0]
    J = lambda x: h.jac(x).toarray()
    cons = []
    if len(eq):
        cons.append(dict(type="eq", fun=lambda x: h.cons(x)[eq] - lc[eq], jac=lambda x: J(x)[eq]))
    if len(lo):
        cons.append(dict(type="ineq", fun=lambda x: h.cons(x)[lo] - lc[lo], jac=lambda x: J(x)[lo]))
    if len(up):
        cons.append(dict(type="ineq", fun=lambda x: uc[up] - h.cons(x)[up], jac=lambda x: -J(x)[up]))
    b = [(None if l == -np.inf else l, None if u == np.inf else u) for l, u in zip(m.lvar, m.uvar)]
    return minimize(h.obj, x0, jac=h.grad, bounds=b, constraints=cons, method="SLSQP",
                    options=dict(ftol=1e-15, maxiter=1000))


def test_known_optima_through_the_gpu_model(built):
    from infiniteexamodels.jl_amd.model import ExaModel
    # test/solve.jl:134-154 — finite parameters, then set_parameter_value + re-solve
    m, (P1, P2) = cases.rosenbrock()
    data = transcribe.ExaMappingData()
    core = transcribe.exa_core(m, data)
    h = HostView(ExaModel(core, device=0))
    x0 = np.array([0.4, 0.4, 0.4, 2.2, 2.2, 2.2])
    f = slsqp(h, x0).fun
    assert abs(f - 306.5) < 1e-6 and abs(f - 306.4999755050365) < 5e-5
    core.set_parameter(data.param_mappings[P1], [90.0])     # ExaModels.set_parameter! → iem_set_parameter
    core.set_parameter(data.param_mappings[P2], [1.3])
    f = slsqp(h, x0).fun
    assert abs(f - 276.265) < 1e-6 and abs(f - 276.26497794903645) < 5e-5
    # test/solve.jl:173-187 — parameter functions
    m, _ = cases.pfun()
    h = HostView(ExaModel(transcribe.exa_core(m), device=0))
    f = slsqp(h, np.full(h.meta.nvar, 1.0)).fun
    assert abs(f - 0.48292223509341475) < 2e-6
    # test/ipopt.jl:160-181 — the warm-start problem, Jacobian AND Hessian from the device
    h = HostView(ExaModel(cases.build_core("ode_5x5"), device=0))
    f = slsqp(h, h.meta.x0.copy()).fun
    assert abs(f - (-12.784599900757165)) < 1e-6
    zero_y = np.zeros(h.meta.ncon)
    con = NonlinearConstraint(h.cons, h.meta.lcon, h.meta.ucon, jac=h.jac, hess=lambda x, v: h.hess(x, v, 0.0))
    res = minimize(h.obj, h.meta.x0.copy(), jac=h.grad, hess=lambda x: h.hess(x, zero_y, 1.0),
                   bounds=Bounds(h.meta.lvar, h.meta.uvar), constraints=[con], method="trust-constr",
                   options=dict(gtol=1e-10, xtol=1e-12, barrier_tol=1e-10, maxiter=3000))
    assert abs(res.fun - (-12.784599900757165)) < 5e-6


def test_quadrotor_100_plumbing(built):
    """Config 1 of BASELINE.json (examples/quadrotor.jl, 100 supports): a few SQP iterations on
    the device model decrease the objective and the constraint violation from the start point."""
    from infiniteexamodels.jl_amd import workloads
    from infiniteexamodels.jl_amd.model import ExaModel
    h = HostView(ExaModel(transcribe.exa_core(workloads.quadrotor(100)), device=0))
    assert (h.meta.nvar, h.meta.ncon) == (2200, 1800)
    x0 = h.meta.x0.copy()
    v0 = np.abs(h.cons(x0)).max()
    assert v0 == pytest.approx(9.8)                      # ∂x6 = u1·cos·cos − 9.8 at x = 0
    # one Gauss-Newton feasibility step with the device Jacobian: min ||c + J dx||
    J = h.jac(x0)
    import scipy.sparse.linalg as spla
    dx = spla.lsqr(J, -h.cons(x0), atol=1e-12, btol=1e-12)[0]
    assert np.abs(h.cons(x0 + dx)).max() < 0.2 * v0


def test_lagrange_newton_solver_in_the_backend_slot(built):
    """``ExaTranscriptionBackend(LagrangeNewtonSolver(), backend = MI355XBackend())``: `optimize()` runs whole Newton
    iterations on the device — the five evaluation calls, the KKT assembly, the chain factorisation (inertia from its pivot
    signs) and solve — and ends at a KKT point with the inertia of a minimiser (re-checked through the oracle on the host);
    models with bounds or inequality rows are refused loudly."""
    import torch
    from infiniteexamodels.jl_amd import lib as iemlib, workloads
    from infiniteexamodels.jl_amd.backend import ExaTranscriptionBackend
    from infiniteexamodels.jl_amd.model import ExaModel, MI355XBackend
    from infiniteexamodels.jl_amd.contrib.newton import LagrangeNewtonSolver
    im = workloads.quadrotor(200, backend=ExaTranscriptionBackend(LagrangeNewtonSolver(tol=1e-8, max_iter=40), backend=MI355XBackend()))
    res = im.optimize()
    assert res.status == "first_order" and res.kkt_residual <= 1e-8 and res.iterations <= 25, res.history
    pos, neg, doubtful = res.history[-2]["inertia"]
    assert (pos, neg, doubtful) == (im.backend.model.meta.nvar, im.backend.model.meta.ncon, 0)      # a minimiser
    # feasibility and stationarity re-checked through the oracle on the host
    from pyoracle import OracleModel
    om = OracleModel(im.backend.core.to_blob())
    x, y = res.solution.cpu().numpy(), res.multipliers.cpu().numpy()
    assert np.abs(om.cons(x) - om.lcon).max() <= 1e-8
    assert np.abs(om.grad(x) + om.jtprod(x, y)).max() <= 1e-7
    assert abs(res.objective - om.obj(x)) <= 1e-9 * max(1.0, abs(om.obj(x)))
    # the result side of the plug point, as the reference's tests read it (test/solve.jl:15-18): statuses, objective,
    # value(x) over the supports, duals of the ODE rows in JuMP's sign
    assert im.termination_status() == "LOCALLY_SOLVED" and im.primal_status() == "FEASIBLE_POINT"
    assert im.objective_value() == res.objective and im.solve_time() > 0.0
    x1 = im.infinite_variables[0]
    v = im.value(x1)
    assert v.shape == (200,) and np.array_equal(v, x[:200]) and im.supports(x1).shape == (200, 1)
    c0 = im.constraints[0]
    con = im.backend.transformation_constraint(c0)
    assert np.array_equal(im.dual(c0), -y[con.offset:con.offset + con.length])
    # the backend's settings reach the solver under Ipopt's option names: a time limit of nothing ends the re-solve at once
    assert im.backend.prev_options == {} or "print_level" not in im.backend.prev_options      # not silent: the default level is not re-sent
    im.set_time_limit_sec(0.0)
    res2 = im.optimize()
    assert im.backend.prev_options["max_wall_time"] == 0.0 and res2.status == "max_time" and im.termination_status() == "TIME_LIMIT"
    im.set_time_limit_sec(None)
    im.set_silent()
    res3 = im.optimize()
    assert im.backend.prev_options == {"max_wall_time": 1.0e20, "print_level": 0} and res3.status == "first_order"
    # a model without a chain (finite parameters only) goes through the dense fallback ... if it is equality-constrained;
    # rosenbrock has inequality rows: refused
    m, _ = cases.rosenbrock()
    gm = ExaModel(transcribe.exa_core(m), device=0)
    with pytest.raises(iemlib.IemError, match="equality-constrained"):
        LagrangeNewtonSolver()(gm)
    gm.close()


def test_lagrange_newton_solver_on_a_maximisation(built):
    """``@objective(m, Max, -f)`` ends at the minimiser of ``f`` with the objective negated and multipliers that make
    ``grad(objective) + J'y`` vanish for the objective as stated."""
    from infiniteexamodels.jl_amd.backend import ExaTranscriptionBackend
    from infiniteexamodels.jl_amd.infinite import InfiniteModel
    from infiniteexamodels.jl_amd.model import MI355XBackend
    from infiniteexamodels.jl_amd.contrib.newton import LagrangeNewtonSolver
    from pyoracle import OracleModel

    def build(sense):
        m = InfiniteModel(ExaTranscriptionBackend(LagrangeNewtonSolver(tol=1e-10), backend=MI355XBackend()))
        t = m.infinite_parameter("t", 0, 1, num_supports=70)
        x, u = m.variable("x", t), m.variable("u", t)
        m.constraint(m.deriv(x, t) == u - x ** 2)
        m.constraint(x(0) == 1.0)
        f = m.integral(x ** 2 + 0.1 * u ** 2 + 0.5 * u, t)
        m.objective(sense, f if sense == "min" else -1.0 * f)
        return m, x

    (mn, xn), (mx, xx) = build("min"), build("max")
    rn, rx = mn.optimize(), mx.optimize()
    assert rn.status == "first_order" and rx.status == "first_order"
    assert abs(mn.objective_value() + mx.objective_value()) <= 1e-9 and np.allclose(mn.value(xn), mx.value(xx), atol=1e-8)
    om = OracleModel(mx.backend.core.to_blob())
    xs, ys = rx.solution.cpu().numpy(), rx.multipliers.cpu().numpy()
    assert not om.minimize and np.abs(om.grad(xs) + om.jtprod(xs, ys)).max() <= 1e-8
    assert np.allclose(mn.dual(mn.constraints[0]), -mx.dual(mx.constraints[0]), atol=1e-7)


@pytest.mark.parametrize("build, max_iterations", [
    (lambda be: __import__("infiniteexamodels.jl_amd.workloads", fromlist=["x"]).quadrotor(2000, backend=be), 10),   # stalled under a residual-only line search
    (lambda be: __import__("infiniteexamodels.jl_amd.workloads", fromlist=["x"]).quadrotor(700, collocation=3, backend=be), 10),
    (lambda be: __import__("infiniteexamodels.jl_amd.workloads", fromlist=["x"]).hovercraft(backend=be), 4),
    (lambda be: __import__("infiniteexamodels.jl_amd.workloads", fromlist=["x"]).hovercraft(21, collocation=4, backend=be), 4),
])
def test_lagrange_newton_solver_on_the_equality_constrained_examples(built, build, max_iterations):
    """the reference's examples without bounds or inequality rows (examples/quadrotor.jl, quadrotor_example.jl,
    hovercraft_example.jl), from their own start values: a KKT point in a handful of iterations, feasibility and
    stationarity re-checked through the oracle"""
    from infiniteexamodels.jl_amd.backend import ExaTranscriptionBackend
    from infiniteexamodels.jl_amd.model import MI355XBackend
    from infiniteexamodels.jl_amd.contrib.newton import LagrangeNewtonSolver
    from pyoracle import OracleModel
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        im = build(ExaTranscriptionBackend(LagrangeNewtonSolver(tol=1e-8, max_iter=40), backend=MI355XBackend()))
        res = im.optimize()
    assert im.termination_status() == "LOCALLY_SOLVED" and res.iterations <= max_iterations, res.history
    om = OracleModel(im.backend.core.to_blob())
    x, y = res.solution.cpu().numpy(), res.multipliers.cpu().numpy()
    assert np.abs(om.cons(x) - om.lcon).max() <= 1e-7 and np.abs(om.grad(x) + om.jtprod(x, y)).max() <= 1e-6


def test_interior_point_solver_reaches_the_references_constants_on_the_device(built):
    """``ExaTranscriptionBackend(InteriorPointSolver(), backend = MI355XBackend())``: evaluation calls, KKT assembly with the
    barrier terms, chain factorisation (inertia) and solves on the device — the constants the reference's tests assert
    (test/solve.jl:146,154,187,206, test/ipopt.jl:180-181) at their tolerance, Ipopt's iteration count on the warm-start
    problem, and the two-stage LP of examples/2stage_example.jl."""
    from infiniteexamodels.jl_amd import workloads
    from infiniteexamodels.jl_amd.backend import ExaTranscriptionBackend
    from infiniteexamodels.jl_amd.contrib.ipm import InteriorPointSolver
    from infiniteexamodels.jl_amd.model import MI355XBackend
    from pyoracle import OracleModel
    mk = lambda **kw: ExaTranscriptionBackend(InteriorPointSolver(**kw), backend=MI355XBackend())
    m, (P1, P2) = cases.rosenbrock()
    m.set_transformation_backend(mk())
    m.optimize()
    assert m.termination_status() == "LOCALLY_SOLVED" and abs(m.objective_value() - 306.4999755050365) < 1e-6
    m.set_parameter_value(P1, 90.0)
    m.set_parameter_value(P2, 1.3)
    m.optimize()
    assert abs(m.objective_value() - 276.26497794903645) < 1e-6
    m, (pf1, pf2) = cases.pfun()
    m.set_transformation_backend(mk())
    m.optimize()
    assert abs(m.objective_value() - 0.48292223509341475) < 1e-6
    m.set_parameter_value(pf1, np.cos)
    m.set_parameter_value(pf2, lambda t, s: np.sin(t) * s + 0.8)
    m.optimize()
    assert abs(m.objective_value() - 0.8155916466182952) < 1e-6
    m = cases.ode_5x5()
    m.set_transformation_backend(mk())
    r = m.optimize()
    assert abs(m.objective_value() - (-12.784599900757165)) < 1e-6 and r.iterations == 8
    m.backend.warmstart_backend_start_values()
    assert m.optimize().iterations < 8
    om7 = workloads.opf(7, backend=mk())                       # ESCAPE34/opf.jl: bounds, inequality rows, a fixed reference angle
    r7 = om7.optimize()
    assert om7.termination_status() == "LOCALLY_SOLVED" and r7.iterations <= 60, (r7.status, r7.iterations, r7.kkt_residual)
    im = workloads.farmer(2000, backend=mk(mu_from_start=True))
    r = im.optimize()
    assert im.termination_status() == "LOCALLY_SOLVED", (r.status, r.iterations, r.kkt_residual)
    om = OracleModel(im.backend.core.to_blob())
    x, y, zL, zU = (a.cpu().numpy() for a in (r.solution, r.multipliers, r.multipliers_L, r.multipliers_U))
    c = om.cons(x)
    tol = lambda b: 1e-7 * np.maximum(1.0, np.abs(np.where(np.isfinite(b), b, 0.0)))      # (bounds are relaxed by 1e-8 relative, as Ipopt does)
    assert (x >= om.lvar - tol(om.lvar)).all() and (c <= om.ucon + tol(om.ucon)).all() and (c >= om.lcon - tol(om.lcon)).all()
    assert np.abs(om.grad(x) + om.jtprod(x, y) - zL + zU).max() <= 1e-6 * max(1.0, np.abs(y).max(), np.abs(zL).max())
