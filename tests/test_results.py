"""The result side of the plug point (``backend.py`` / ``contrib/results.py``): what the reference's tests read back after
``optimize!`` — ``objective_value``, ``value(...)``, ``dual(...)``, statuses, supports — and the option diffing of a
re-solve.  CPU only: the "solver" is SciPy's SLSQP on the oracle's evaluator (a stand-in for IpoptSolver of
/root/reference/test/solve.jl), so the backend is built without a device (``backend = None``)."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import minimize

import cases
from infiniteexamodels.jl_amd.contrib import results as R
from infiniteexamodels.jl_amd.backend import ExaTranscriptionBackend
from infiniteexamodels.jl_amd.infinite import DomainRestriction, InfiniteModel, OrthogonalCollocation
from pyoracle import OracleModel


class OracleSLSQP:
    """``solver(model, x0, y0, **options) -> result`` on the oracle; multipliers from the KKT conditions at the solution
    (least squares over the active rows and bounds), NLPModels sign: grad f + J'y - zL + zU = 0."""
    option_convention = "ipopt"

    def __init__(self, backend_ref, x0=None):
        self.be, self.x0, self.calls = backend_ref, x0, []

    def __call__(self, model, x0, y0, **options):
        self.calls.append(("initial", dict(options)))
        return self._solve(x0)

    def resolve(self, model, x0, y0, **options):
        self.calls.append(("resolve", dict(options)))
        return self._solve(x0)

    def _solve(self, x0):
        om = OracleModel(self.be[0].core.to_blob())
        jr, jc = om.jac_structure()
        J = lambda x: sp.coo_matrix((om.jac_coord(x), (jr, jc)), shape=(om.ncon, om.nvar)).toarray()
        lc, uc = om.lcon, om.ucon
        cons = []
        eq = np.nonzero(lc == uc)[0]
        lo = np.nonzero((lc > -np.inf) & (lc != uc))[0]
        up = np.nonzero((uc < np.inf) & (lc != uc))[0]
        if len(eq):
            cons.append(dict(type="eq", fun=lambda x: om.cons(x)[eq] - lc[eq], jac=lambda x: J(x)[eq]))
        if len(lo):
            cons.append(dict(type="ineq", fun=lambda x: om.cons(x)[lo] - lc[lo], jac=lambda x: J(x)[lo]))
        if len(up):
            cons.append(dict(type="ineq", fun=lambda x: uc[up] - om.cons(x)[up], jac=lambda x: -J(x)[up]))
        b = [(None if l == -np.inf else l, None if u == np.inf else u) for l, u in zip(om.lvar, om.uvar)]
        start = np.asarray(self.x0 if self.x0 is not None else x0, dtype=float)
        res = minimize(om.obj, start, jac=om.grad, bounds=b, constraints=cons, method="SLSQP", options=dict(ftol=1e-15, maxiter=1000))
        x = res.x
        c = om.cons(x)
        act = np.nonzero((np.abs(c - lc) < 1e-7) | (np.abs(c - uc) < 1e-7))[0]
        bl, bu = np.nonzero(np.abs(x - om.lvar) < 1e-9)[0], np.nonzero(np.abs(x - om.uvar) < 1e-9)[0]
        A = np.hstack([J(x)[act].T, -np.eye(om.nvar)[:, bl], np.eye(om.nvar)[:, bu]])
        mult = np.linalg.lstsq(A, -om.grad(x), rcond=None)[0] if A.shape[1] else np.zeros(0)
        y, zL, zU = np.zeros(om.ncon), np.zeros(om.nvar), np.zeros(om.nvar)
        y[act] = mult[:len(act)]
        zL[bl] = mult[len(act):len(act) + len(bl)]
        zU[bu] = mult[len(act) + len(bl):]
        kkt = om.grad(x) + J(x).T @ y - zL + zU          # (SLSQP at ftol 1e-15 often stops on its line search: judge the point itself)
        return type("Stats", (), dict(solution=x, multipliers=y, multipliers_L=zL, multipliers_U=zU, objective=res.fun,
                                      status="first_order" if np.abs(kkt).max() < 1e-5 else "max_iter", elapsed_time=0.0))()


def attach(m, x0=None):
    holder = []
    be = ExaTranscriptionBackend(None)
    holder.append(be)
    be.set_optimizer(OracleSLSQP(holder, x0))
    m.set_transformation_backend(be)
    return be


def test_finite_parameters_resolve_and_queries(built):
    """/root/reference/test/solve.jl:134-162"""
    m, (P1, P2) = cases.rosenbrock()
    be = attach(m, x0=np.array([0.4, 0.4, 0.4, 2.2, 2.2, 2.2]))
    assert be.termination_status() == "OPTIMIZE_NOT_CALLED" and be.primal_status() == "NO_SOLUTION" and be.result_count() == 0
    assert be.raw_status() == "optimize not called"
    with pytest.raises(RuntimeError, match="No solution available"):
        m.objective_value()
    m.set_silent()
    m.optimize()
    assert abs(m.objective_value() - 306.4999755050365) < 5e-5          # :146 (Ipopt's value carries its bound relaxation)
    assert m.value(P1) == 100.0 and m.value(P2) == 1.0                   # :147-148
    assert m.termination_status() == "LOCALLY_SOLVED" and m.primal_status() == "FEASIBLE_POINT" and be.result_count() == 1
    assert be.raw_status() == "first_order" and m.solve_time() >= 0.0
    m.set_parameter_value(P1, 90.0)                                      # :150-151: theta in place, no rebuild
    m.set_parameter_value(P2, 1.3)
    assert m.transformation_backend_ready()
    m.optimize()
    assert abs(m.objective_value() - 276.26497794903645) < 5e-5          # :154
    assert m.value(P1) == 90.0 and m.value(P2) == 1.3
    # every support solves x = (0.5, 2): value(x[i]) has the shape of the supports
    x1, x2 = m.infinite_variables[:2]
    assert m.value(x1).shape == (3,) and np.allclose(m.value(x1), 0.5, atol=1e-6) and np.allclose(m.value(x2), 2.0, atol=1e-5)
    assert m.supports(x1).shape == (3, 1) and np.allclose(m.supports(x1)[:, 0], [0.0, 0.5, 1.0])
    # the first solve went to the solver whole, the second as a resolve with only what changed (nothing: theta is not an option)
    kinds = [k for k, _ in be.solver.calls]
    assert kinds == ["initial", "resolve"]
    assert be.solver.calls[0][1] == {"print_level": 0} and be.solver.calls[1][1] == {}


def test_duals_carry_jump_sign(built):
    """min int (x - 2)^2 dt  s.t.  x <= 1 (a constraint) and y >= 3 (a variable bound), y enters as (y - 1)^2:
    duals are -(NLPModels multipliers) (src/infiniteopt_backend.jl:490-508), bound duals mL - mU clipped by sense."""
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, num_supports=5)
    x = m.variable("x", t)
    y = m.variable("y", t, lb=3.0)
    c = m.constraint(x <= 1.0, name="c")
    m.objective("min", m.integral((x - 2.0) ** 2 + (y - 1.0) ** 2, t))
    attach(m)
    m.optimize()
    w = np.array([0.125, 0.25, 0.25, 0.25, 0.125])                       # trapezoid weights
    assert np.allclose(m.value(x), 1.0, atol=1e-6) and np.allclose(m.value(y), 3.0, atol=1e-6)
    assert np.allclose(m.dual(c), -2.0 * w, atol=1e-5)                    # d obj / d rhs of a <= row: non-positive
    assert np.allclose(m.dual((y, "lower")), 4.0 * w, atol=1e-5)          # a lower bound: non-negative
    assert np.allclose(m.dual((y, "upper")), 0.0)
    assert m.supports(c).shape == (5, 1)


def test_parameter_functions_values(built):
    """/root/reference/test/solve.jl:173-208: value(pf) is the slab of theta the model evaluates with."""
    m, (pf1, pf2) = cases.pfun()
    attach(m)
    m.optimize()
    assert abs(m.objective_value() - 0.48292223509341475) < 2e-6
    ts, ss = np.array([0.0, 0.5, 1.0]), np.array([2.0, 2.5, 3.0])
    assert np.array_equal(m.value(pf1), np.sin(ts))
    assert m.value(pf2).shape == (3, 3) and np.allclose(m.value(pf2), np.sin(ts)[:, None] * ss[None, :] + 0.2, rtol=0, atol=1e-15)
    m.set_parameter_value(pf1, np.cos)
    m.set_parameter_value(pf2, lambda t, s: np.sin(t) * s + 0.8)              # newpf2 of :199-200
    assert m.transformation_backend_ready()
    m.optimize()
    assert abs(m.objective_value() - 0.8155916466182952) < 2e-6
    assert np.array_equal(m.value(pf1), np.cos(ts))
    z = next(v for v in m.infinite_variables if v.name == "z")
    assert m.value(z).shape == (3, 3) and m.supports(z).shape == (3, 3, 2)
    assert np.allclose(m.supports(z)[1, 2], [0.5, 3.0])                   # first parameter along the first axis


def test_labels_drop_internal_collocation_nodes(built):
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, num_supports=4)
    m.set_derivative_method(t, OrthogonalCollocation(3))
    x = m.variable("x", t, start=1.0)
    dx = m.deriv(x, t)
    c = m.constraint(dx == -x, name="ode")
    m.constraint(x(0) == 1.0)
    m.objective("min", m.integral(x ** 2, t))
    attach(m)
    m.optimize()
    pub, allv, inner = m.value(x), m.value(x, "all"), m.value(x, "internal")
    assert pub.shape == (4,) and allv.shape == (7,) and inner.shape == (3,)     # one internal node per interval
    assert np.array_equal(allv[::2], pub) and np.array_equal(allv[1::2], inner)
    assert m.supports(x).shape == (4, 1) and m.supports(x, "all").shape == (7, 1)
    assert np.allclose(m.supports(x)[:, 0], np.linspace(0, 1, 4))
    assert np.allclose(pub, np.exp(-np.linspace(0, 1, 4)), atol=2e-3)             # x' = -x, x(0) = 1
    assert m.dual(c).shape == (4,) and m.dual(c, "all").shape == (7,)


def test_restricted_constraint_supports(built):
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, num_supports=5)
    x = m.variable("x", t)
    c = m.constraint(x >= 1.0, restriction=DomainRestriction(lambda t: t >= 0.5, t), name="late")
    m.objective("min", m.integral(x ** 2, t))
    attach(m)
    m.optimize()
    assert np.allclose(m.supports(c)[:, 0], [0.5, 0.75, 1.0])                     # test/transcription.jl:217-style count
    assert m.dual(c).shape == (3,) and np.all(m.dual(c) >= -1e-9)                # >= rows: non-negative duals
    assert np.allclose(m.value(x), [0, 0, 1, 1, 1], atol=1e-6)


def test_option_diffing_matches_the_extensions():
    """ext/InfiniteExaModelsIpopt.jl:10-39 step by step"""
    be = type("B", (), dict(prev_options={}, silent=False, time_limit=float("nan")))()
    conv = R.OPTION_CONVENTIONS["ipopt"]
    assert R.process_options({"tol": 1e-8}, be, conv) == {"tol": 1e-8}
    assert R.process_options({"tol": 1e-8}, be, conv) == {}                      # unchanged: nothing to pass
    assert R.process_options({"tol": 1e-6}, be, conv) == {"tol": 1e-6}
    be.silent = True
    assert R.process_options({"tol": 1e-6}, be, conv) == {"print_level": 0}
    assert R.process_options({"tol": 1e-6}, be, conv) == {}
    be.silent = False                                                            # restored to the solver's default
    assert R.process_options({"tol": 1e-6}, be, conv) == {"print_level": 5}
    be.time_limit = 10.0
    assert R.process_options({"tol": 1e-6}, be, conv) == {"max_wall_time": 10.0}
    be.time_limit = float("nan")
    assert R.process_options({"tol": 1e-6}, be, conv) == {"max_wall_time": 1.0e20}
    assert R.process_options({"tol": 1e-6, "max_wall_time": 3.0}, be, conv) == {"max_wall_time": 3.0}
    # MadNLP's conventions (ext/InfiniteExaModelsMadNLP.jl:6-8)
    be2 = type("B", (), dict(prev_options={}, silent=True, time_limit=5.0))()
    assert R.process_options({}, be2, R.OPTION_CONVENTIONS["madnlp"]) == {"print_level": "ERROR", "max_wall_time": 5.0}


def test_attributes_and_status_tables():
    be = ExaTranscriptionBackend(None)
    assert be.get_attribute("solver_name") == "No solver attached" and be.get_attribute("time_limit_sec") is None
    be.set_attribute("tol", 1e-4)
    assert be.get_attribute("tol") == 1e-4
    with pytest.raises(KeyError, match="not found"):
        be.get_attribute("missing")
    be.set_time_limit_sec(7)
    assert be.get_attribute("time_limit_sec") == 7.0
    be.set_time_limit_sec(None)
    assert be.get_attribute("time_limit_sec") is None
    be.set_optimizer(len, linear_solver="ma27")                                   # previous settings dropped (:233-241)
    assert be.options == {"linear_solver": "ma27"} and be.get_attribute("solver_name") == "len"
    assert R.translate_termination_status(None, "max_iter") == "ITERATION_LIMIT"
    assert R.translate_termination_status(None, "something else") == "OTHER_ERROR"
    assert R.translate_result_status(None, "acceptable") == "NEARLY_FEASIBLE_POINT"
    assert R.translate_result_status(None, "max_iter") == "UNKNOWN_RESULT_STATUS"
    madnlp = type("S", (), dict(termination_statuses=R.MADNLP_TERMINATION, result_statuses=R.MADNLP_RESULT))()
    assert R.translate_termination_status(madnlp, "INFEASIBLE_PROBLEM_DETECTED") == "LOCALLY_INFEASIBLE"
    assert R.translate_result_status(madnlp, "SOLVE_SUCCEEDED") == "FEASIBLE_POINT"


def _prev(be):
    return dict(be.prev_options)


@pytest.mark.parametrize("conv, SILENT, DEFAULT, USER, WALL", [("ipopt", 0, 5, 3, 1.0e20), ("madnlp", "ERROR", "INFO", "WARN", 1.0e6)])
def test_option_updates_replayed_from_the_reference(built, conv, SILENT, DEFAULT, USER, WALL):
    """/root/reference/test/ipopt.jl:2-158 and test/madnlp.jl:2-165 ("option updates 1 / 2"): the dictionaries the
    reference asserts for ``prev_options`` after every solve, with SLSQP on the oracle standing in for the solver (its
    optimum: ``test/ipopt.jl:18``) under that solver's option conventions."""
    # --- option updates 1 (:2-55)
    m = cases.ode_5x5()
    be = attach(m)
    be.solver.option_convention = conv
    m.set_silent()
    m.set_time_limit_sec(120.0)
    assert be.silent is True and be.time_limit == 120.0
    m.optimize()
    assert abs(m.objective_value() - (-12.784599900757165)) < 2e-6
    assert be.options == {} and _prev(be) == {"print_level": SILENT, "max_wall_time": 120.0}
    m.set_silent(False)                                   # unset_silent
    m.set_time_limit_sec(200.0)
    for k, v in (("max_iter", 50), ("mu_init", 1e-2), ("tol", 1e-6)):
        be.set_attribute(k, v)                            # set_optimizer_attribute
    assert be.results is not None                         # changing options does not wipe the results
    m.optimize()
    assert be.options == {"max_iter": 50, "mu_init": 1e-2, "tol": 1e-6}
    assert _prev(be) == {"max_iter": 50, "mu_init": 1e-2, "tol": 1e-6, "print_level": DEFAULT, "max_wall_time": 200.0}
    assert [k for k, _ in be.solver.calls] == ["initial", "resolve"]
    assert be.solver.calls[1][1] == {"max_iter": 50, "mu_init": 1e-2, "tol": 1e-6, "print_level": DEFAULT, "max_wall_time": 200.0}
    # --- option updates 2 (:57-158)
    m = cases.ode_5x5()
    be = attach(m)
    be.solver.option_convention = conv
    m.set_time_limit_sec(120.0)
    for k, v in (("max_iter", 50), ("mu_init", 1e-2), ("tol", 1e-6)):
        be.set_attribute(k, v)
    m.optimize()
    assert _prev(be) == {"max_iter": 50, "mu_init": 1e-2, "tol": 1e-6, "max_wall_time": 120.0}
    be.set_attribute("print_level", USER)
    m.set_time_limit_sec(None)                            # unset_time_limit_sec
    assert np.isnan(be.time_limit)
    m.optimize()
    assert _prev(be) == {"print_level": USER, "max_wall_time": WALL, "tol": 1e-6, "mu_init": 1e-2, "max_iter": 50}
    assert be.options == {"max_iter": 50, "mu_init": 1e-2, "tol": 1e-6, "print_level": USER}
    m.set_silent()
    m.set_time_limit_sec(150.0)
    m.optimize()
    assert be.options == {"max_iter": 50, "mu_init": 1e-2, "tol": 1e-6, "print_level": USER}
    assert _prev(be) == {"max_iter": 50, "mu_init": 1e-2, "tol": 1e-6, "max_wall_time": 150.0, "print_level": SILENT}
    m.set_silent(False)                                   # the print level set before comes back
    m.optimize()
    assert _prev(be) == {"print_level": USER, "max_wall_time": 150.0, "tol": 1e-6, "mu_init": 1e-2, "max_iter": 50}
    assert be.solver.calls[-1][1] == {"print_level": USER}


def test_warm_start_and_rebuild_keep_the_settings(built):
    """/root/reference/test/ipopt.jl:159-203 (warm start) and :205-221 (a rebuild re-sends the silent setting, #26)"""
    m = cases.ode_5x5()
    be = attach(m)
    with pytest.warns(UserWarning, match="No previous solution values found"):
        be.warmstart_backend_start_values()
    m.optimize()
    expected = np.zeros(51)
    expected[0] = 10.0
    assert np.array_equal(be.core.x0, expected)                                   # :183-185
    be.warmstart_backend_start_values()
    assert np.array_equal(be.core.x0, be.results.solution)                        # :190-191
    seen = []
    inner = be.solver._solve
    be.solver._solve = lambda x0: (seen.append(np.array(x0)), inner(x0))[1]
    m.optimize()
    assert np.array_equal(seen[-1], be.results.solution) or np.allclose(seen[-1], be.results.solution, atol=1e-6)   # started from the solution
    assert abs(m.objective_value() - (-12.784599900757165)) < 2e-6
    # a rebuild: prev_options are emptied with the backend, so the silent setting travels again
    m2 = InfiniteModel()
    t = m2.infinite_parameter("t", 0, 1, num_supports=5)
    y = m2.variable("y", t, lb=0, start=1.0)
    z = m2.finite_parameter("z", 10.0)
    m2.objective("min", m2.integral(y ** 2 + 2 * z, t))
    m2.constraint(y + z <= 42 + t)
    be2 = attach(m2)
    m2.set_silent()
    m2.optimize()
    assert be2.solver.calls[-1] == ("initial", {"print_level": 0})
    m2.variable("w", t)                                   # a change the backend cannot take in place
    assert not m2.transformation_backend_ready()
    m2.objective("min", m2.integral(y ** 2 + 2 * z, t))
    m2.optimize()
    assert be2.solver.calls[-1] == ("initial", {"print_level": 0})                # still silent, and a fresh solver state


@pytest.mark.parametrize("maker, nt", [(cases.test_problem_1, 5), (cases.test_problem_1_oc3, 9)])
def test_values_of_variables_derivatives_and_restricted_rows(built, maker, nt):
    """/root/reference/test/solve.jl:2-44: ``value(y)``, ``value(z)``, ``value(∂(y, t))`` after a solve — the reference
    compares them with InfiniteOpt's own transcription; here they are checked against what the transcription promises:
    shapes over the (public) supports, the derivative values satisfying the ODE row they appear in, the restricted row
    holding on the supports it admits."""
    m = maker()
    attach(m)
    m.optimize()
    y = next(v for v in m.infinite_variables if v.name == "y")
    z = m.finite_variables[0]
    dy = m.derivatives[0]
    Y, Z, DY = m.value(y), m.value(z), m.value(dy)
    assert Y.shape == (5, 5) and DY.shape == (5, 5) and np.isscalar(Z)             # public supports only
    assert m.value(y, "all").shape == (nt, 5) and m.supports(y, "all").shape == (nt, 5, 2)
    assert np.allclose(DY, np.sin(Y) + Z + 1.2, atol=1e-6)                          # dy/dt == sin(y) + z + 1.2
    assert (Y >= -1e-9).all()
    c_ode, c_restricted = m.constraints[0], m.constraints[1]
    assert m.dual(c_ode, "all").shape == (nt * 5,) or m.dual(c_ode, "all").shape == (nt, 5)
    supp = m.supports(c_restricted, "all")
    assert supp.shape[1] == 2 and ((supp[:, 0] >= 0) & (supp[:, 0] <= 0.5)).all()
    assert m.dual(c_restricted, "all").shape == (supp.shape[0],)
    ts = m.supports(y)[:, 0, 0]
    early = ts <= 0.5
    assert (Y[early] + Z <= 42 + ts[early][:, None] + 1e-7).all()
