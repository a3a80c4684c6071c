"""The oracle against an INDEPENDENT derivative source: torch float64 autograd over a
separate Python evaluator of the recorded expression trees (tests/helpers.py).  COO output
is scattered to dense (duplicates summed; Hessian mirrored from the lower triangle)."""
import numpy as np
import pytest

import cases
from helpers import TorchModel, coo_to_dense, lower_to_full
from pyoracle import OracleModel

SMALL = ["quadrotor_5", "quadrotor_oc3_40", "pandemic_20x3", "farmer_5", "ode_5x5", "test_problem_1", "rosenbrock", "pfun",
         "irregular", "hovercraft_oc4", "three_node_50", "kinetic_20", "test_problem_1_oc3", "test_problem_2_obj2",
         "test_problem_2_obj3", "test_problem_2_obj4", "pfun_full"]


@pytest.mark.parametrize("name", SMALL)
def test_oracle_matches_autograd(name, built):
    core = cases.build_core(name)
    om = OracleModel(core.to_blob())
    x, y = cases.eval_point_for(name, om)
    f, c, g, J, H = TorchModel(core).dense(x, y, 0.7)
    assert abs(om.obj(x) - f) <= 1e-12 * max(1.0, abs(f))
    np.testing.assert_allclose(om.cons(x), c, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(om.grad(x), g, rtol=1e-12, atol=1e-12)
    r, cc = om.jac_structure()
    np.testing.assert_allclose(coo_to_dense(r, cc, om.jac_coord(x), (om.ncon, om.nvar)), J, rtol=1e-12, atol=1e-12)
    r, cc = om.hess_structure()
    assert (r >= cc).all(), "Hessian structure must be lower triangular"
    Ho = lower_to_full(coo_to_dense(r, cc, om.hess_coord(x, y, 0.7), (om.nvar, om.nvar)))
    np.testing.assert_allclose(Ho, H, rtol=1e-11, atol=1e-11 * max(1.0, np.abs(H).max()))
    # matrix-free products against the dense autograd matrices
    rng = np.random.default_rng(9)
    v, vc = rng.standard_normal(om.nvar), rng.standard_normal(om.ncon)
    np.testing.assert_allclose(om.jprod(x, v), J @ v, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(om.jtprod(x, vc), J.T @ vc, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(om.hprod(x, y, v, 0.7), H @ v, rtol=1e-10, atol=1e-10 * max(1.0, np.abs(H).max()))


def test_operator_zoo_first_and_second_derivatives(built):
    """Every operator of src/operators.jl:3-44: f, f', f'' of the oracle's table vs autograd."""
    core = cases.build_core("operator_zoo")
    om = OracleModel(core.to_blob())
    x, y = cases.eval_point_for("operator_zoo", om)
    f, c, g, J, H = TorchModel(core).dense(x, y, 1.3)
    np.testing.assert_allclose(om.cons(x), c, rtol=1e-12, atol=1e-12)
    r, cc = om.jac_structure()
    np.testing.assert_allclose(coo_to_dense(r, cc, om.jac_coord(x), (om.ncon, om.nvar)), J, rtol=1e-10, atol=1e-10)
    r, cc = om.hess_structure()
    Ho = lower_to_full(coo_to_dense(r, cc, om.hess_coord(x, y, 1.3), (om.nvar, om.nvar)))
    np.testing.assert_allclose(Ho, H, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(H).max()))


def test_csch_maps_to_csc_like_the_reference(built):
    """operators.jl:40 maps :csch to csc; the transcriber reproduces that table entry."""
    from infiniteexamodels.jl_amd import nodes as N
    from infiniteexamodels.jl_amd.operators import nl_op
    v = N.Var(1)
    assert nl_op("csch")(v) == N.FUNCS["csc"](v)
    with pytest.raises(KeyError, match="does not support the nonlinear operator `asinh`"):
        nl_op("asinh")


def test_structure_layout_rules(built):
    """Layout rules restated from ExaModels: rows = o0+k, block position o1 + o1step*k + s,
    Hessian rows >= cols, offsets are running counters in call order."""
    core = cases.build_core("quadrotor_100")
    om = OracleModel(core.to_blob())
    S = 100
    assert (om.nvar, om.ncon, om.npar) == (22 * S, 18 * S, 3 * S)
    assert om.nnzj == 62 * S - 18
    infos = [om.template_info(i) for i in range(om.n_templates)]
    assert [t["n_items"] for t in infos] == [1] * 9 + [S] * 9 + [S - 1] * 9 + [S]
    assert [t["o1step"] for t in infos[:27]] == [1] * 9 + [2, 5, 2, 5, 2, 4, 5, 4, 6] + [3] * 9
    o0 = o1 = o2 = 0
    for t in infos:
        assert t["o2"] == o2
        o2 += t["n_items"] * t["o2step"]
        if t["kind"] == 1:
            assert (t["o0"], t["o1"]) == (o0, o1)
            o0 += t["n_items"]
            o1 += t["n_items"] * t["o1step"]
    r, c = om.jac_structure(base=1)
    assert r.min() == 1 and r.max() == om.ncon and c.min() >= 1 and c.max() <= om.nvar
    # first nine rows: x_k(0) == 0 → column = first entry of slab k (transform.jl:259-270)
    assert list(c[:9]) == [1 + k * S for k in range(9)]
    # finite-difference row of x_1 at i = 2: columns ∂x1[2], x1[2], x1[1] in visit order
    t18 = infos[18]
    blk = slice(t18["o1"], t18["o1"] + 3)
    assert list(c[blk]) == [13 * S + 2, 2, 1]
