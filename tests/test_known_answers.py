"""Solver-level known answers of the reference's own test-suite, reproduced with scipy on
the oracle's evaluator (objective, gradient, sparse Jacobian and Lagrangian Hessian all come
from oracle/): a wrong transcription, Jacobian or Hessian breaks these constants.

  306.4999755050365 / 276.26497794903645     /root/reference/test/solve.jl:146,154
  0.48292223509341475 / 0.8155916466182952   /root/reference/test/solve.jl:187,206
  -12.784599900757165                        /root/reference/test/ipopt.jl:181
The reference asserts them at atol = 1e-6 (test/solve.jl:1); Ipopt stops at its own
tolerance, so the constants themselves carry ~1e-7 of solver noise.
"""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import Bounds, NonlinearConstraint, minimize

import cases
from infiniteexamodels.jl_amd import transcribe
from pyoracle import OracleModel

TOL = 2e-6


def slsqp(om: OracleModel, x0=None):
    """Active-set SQP (exact at bounds) on the oracle's obj / grad / cons / Jacobian."""
    jr, jc = om.jac_structure()
    J = lambda x: sp.coo_matrix((om.jac_coord(x), (jr, jc)), shape=(om.ncon, om.nvar)).toarray()
    lc, uc = om.lcon, om.ucon
    eq = np.nonzero(lc == uc)[0]
    lo = np.nonzero((lc > -np.inf) & (lc != uc))[0]
    up = np.nonzero((uc < np.inf) & (lc != uc))[0]
    cons = []
    if len(eq):
        cons.append(dict(type="eq", fun=lambda x: om.cons(x)[eq] - lc[eq], jac=lambda x: J(x)[eq]))
    if len(lo):
        cons.append(dict(type="ineq", fun=lambda x: om.cons(x)[lo] - lc[lo], jac=lambda x: J(x)[lo]))
    if len(up):
        cons.append(dict(type="ineq", fun=lambda x: uc[up] - om.cons(x)[up], jac=lambda x: -J(x)[up]))
    b = [(None if l == -np.inf else l, None if u == np.inf else u) for l, u in zip(om.lvar, om.uvar)]
    return minimize(om.obj, om.x0 if x0 is None else x0, jac=om.grad, bounds=b, constraints=cons,
                    method="SLSQP", options=dict(ftol=1e-15, maxiter=1000))


def solve(om: OracleModel, x0=None):
    n, m = om.nvar, om.ncon
    jr, jc = om.jac_structure()
    hr, hc = om.hess_structure()

    def full_h(vals):
        L = sp.coo_matrix((vals, (hr, hc)), shape=(n, n)).tocsr()
        return (L + L.T - sp.diags(L.diagonal())).tocsr()

    zero_y = np.zeros(m)
    cons = NonlinearConstraint(
        om.cons, om.lcon, om.ucon,
        jac=lambda x: sp.coo_matrix((om.jac_coord(x), (jr, jc)), shape=(m, n)).tocsr(),
        hess=lambda x, v: full_h(om.hess_coord(x, v, 0.0)))
    res = minimize(om.obj, om.x0 if x0 is None else x0, jac=om.grad,
                   hess=lambda x: full_h(om.hess_coord(x, zero_y, 1.0)),
                   bounds=Bounds(om.lvar, om.uvar), constraints=[cons], method="trust-constr",
                   options=dict(gtol=1e-10, xtol=1e-12, barrier_tol=1e-10, maxiter=3000))
    return res


def test_rosenbrock_like_with_finite_parameters(built):
    m, (P1, P2) = cases.rosenbrock()
    data = transcribe.ExaMappingData()
    core = transcribe.exa_core(m, data)
    om = OracleModel(core.to_blob())
    x0 = np.array([0.4, 0.4, 0.4, 2.2, 2.2, 2.2])
    f = slsqp(om, x0).fun
    assert abs(f - 306.5) < 1e-6                     # analytic: every support solves x = (0.5, 2)
    assert abs(f - 306.4999755050365) < 5e-5         # the reference's Ipopt value (bound relaxation noise)
    # set_parameter_value(p1, 90.0); set_parameter_value(p2, 1.3)  →  ExaModels.set_parameter!
    om.set_parameter(data.param_mappings[P1].offset, [90.0])
    om.set_parameter(data.param_mappings[P2].offset, [1.3])
    assert list(om.theta) == [90.0, 1.3]
    f = slsqp(om, x0).fun
    assert abs(f - 276.265) < 1e-6
    assert abs(f - 276.26497794903645) < 5e-5


def test_parameter_function_problem_and_update(built):
    m, (pf1, pf2) = cases.pfun()
    data = transcribe.ExaMappingData()
    core = transcribe.exa_core(m, data)
    om = OracleModel(core.to_blob())
    x0 = np.full(om.nvar, 1.0)
    f = slsqp(om, x0).fun
    assert abs(f - 0.4829222187) < 1e-9              # closed form Σ c_t·0.5·(3 sin t + k)·pf1(t)
    assert abs(f - 0.48292223509341475) < TOL
    # set_parameter_value(pf1, cos); set_parameter_value(pf2, newpf2)  (test/solve.jl:196-204)
    t, s = np.array([0.0, 0.5, 1.0]), np.array([2.0, 2.5, 3.0])
    new1 = np.cos(t)
    new2 = np.array([np.sin(a) * b + 0.8 for b in s for a in t])
    om.set_parameter(data.param_mappings[pf1].offset, new1)
    om.set_parameter(data.param_mappings[pf2].offset, new2)
    expected2 = [0.8, 1.758851077208406, 2.4829419696157933, 0.8, 1.9985638465105076, 2.9036774620197416,
                 0.8, 2.238276615812609, 3.324412954423689]
    p2 = data.param_mappings[pf2]
    np.testing.assert_array_equal(om.theta[p2.offset:p2.offset + 9], expected2)
    f = slsqp(om, x0).fun
    assert abs(f - 0.8155916298) < 1e-9
    assert abs(f - 0.8155916466182952) < TOL


def test_ode_5x5_warmstart_problem(built):
    core = cases.build_core("ode_5x5")
    om = OracleModel(core.to_blob())
    assert (om.nvar, om.ncon) == (51, 70)
    f = slsqp(om).fun
    assert abs(f - (-12.784599900757165)) < 1e-6, f          # test/ipopt.jl:181
    assert abs(f - (-12.784599867885884)) < 1e-6, f          # test/madnlp.jl:42
    # interior point with the exact sparse Jacobian AND Lagrangian Hessian of the oracle
    res = solve(om)
    assert abs(res.fun - (-12.784599900757165)) < 5e-6, res.fun
