"""Host mirror of src/transform.jl against the reference's own structural assertions
(/root/reference/test/transcription.jl, test/ipopt.jl:183-186, test/solve.jl:190-204)."""
import numpy as np
import pytest

import cases
from infiniteexamodels.jl_amd import infinite as io
from infiniteexamodels.jl_amd import transcribe
from infiniteexamodels.jl_amd.core import ExaCore, T_CON, T_OBJ
from infiniteexamodels.jl_amd.infinite import DomainRestriction, InfiniteModel
from infiniteexamodels.jl_amd.nodes import Var


def test_mapping_initializers():
    """test/transcription.jl:1-89 — variable blocks, function-valued bounds, semi-infinite
    and point overrides."""
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, num_supports=5)
    x = m.infinite_parameter("x", -1, 1, num_supports=5)   # 3 supports + collocation nodes in the reference
    y = m.variable("y", t, lb=np.cos, ub=1)
    q = m.variable("q", t, x, fix=42)
    w = m.variable("w", x, lb=2, ub=np.sin, start=np.cos)
    y0, y1 = y(0), y(1)
    y0.info.start = 0.5
    y1.info.lb, y1.info.ub = float("nan"), 0.8      # delete_lower_bound + set_upper_bound
    q0, q1 = q(0, x), q(1, x)
    q0.info.start = 10
    q1.info.fix = 5
    d1 = m.deriv(y, t)
    z = m.variable("z", start=10)
    c, data = ExaCore(), transcribe.ExaMappingData()
    transcribe._build_base_iterators(data, m)
    assert len(data.base_itrs) == 2
    transcribe._add_finite_variables(c, data, m)
    v = data.finvar_mappings[z]
    assert v == Var(1) and c.x0[0] == 10 and c.lvar[0] == -np.inf and c.uvar[0] == np.inf   # :29-36
    transcribe._add_infinite_variables(c, data, m)
    yv = data.infvar_mappings[y]
    assert yv.length == 5
    np.testing.assert_array_equal(c.lvar[yv.offset:yv.offset + 5], np.cos(np.linspace(0, 1, 5)))   # :45
    np.testing.assert_array_equal(c.uvar[yv.offset:yv.offset + 5], np.ones(5))
    qv = data.infvar_mappings[q]
    assert qv.length == 25
    assert (c.lvar[qv.offset:qv.offset + 25] == 42).all() and (c.uvar[qv.offset:qv.offset + 25] == 42).all()
    wv = data.infvar_mappings[w]
    np.testing.assert_array_equal(c.lvar[wv.offset:wv.offset + 5], np.full(5, 2.0))
    np.testing.assert_array_equal(c.uvar[wv.offset:wv.offset + 5], np.sin(np.linspace(-1, 1, 5)))     # :57
    np.testing.assert_array_equal(c.x0[wv.offset:wv.offset + 5], np.cos(np.linspace(-1, 1, 5)))
    assert data.infvar_mappings[d1].length == 5
    transcribe._add_semi_infinite_variables(c, data, m)
    assert len(data.semivar_info) == 2
    assert c.x0[qv[1, 2].i - 1] == 10                                     # :72
    assert c.lvar[qv[5, 3].i - 1] == 5 and c.uvar[qv[5, 4].i - 1] == 5    # :74-75
    assert c.x0[qv[5, 2].i - 1] == 0
    transcribe._add_point_variables(c, data, m)
    assert len(data.finvar_mappings) == 3
    p0 = data.finvar_mappings[y0]
    assert c.x0[p0.i - 1] == 0.5 and c.lvar[p0.i - 1] == np.cos(0)        # :82-83
    p1 = data.finvar_mappings[y1]
    assert c.lvar[p1.i - 1] == -np.inf and c.uvar[p1.i - 1] == 0.8        # :86-87


def test_finite_parameters_go_to_theta():
    """test/transcription.jl:91-127."""
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, num_supports=5)
    x = m.finite_parameter("x", 42)
    ys = [m.finite_parameter(f"y[{i}]", v) for i, v in enumerate([20, 30])]
    v = m.variable("v", t, lb=0, ub=100)
    data = transcribe.ExaMappingData()
    core = transcribe.exa_core(m, data)
    assert len(data.param_mappings) == 3 and core.npar == 3
    assert core.theta[data.param_mappings[x].offset] == 42
    assert [core.theta[data.param_mappings[p].offset] for p in ys] == [20, 30]


def test_parameter_functions_column_major():
    """test/transcription.jl:129-175: θ holds pf values, first parameter fastest."""
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, num_supports=5)
    s = m.infinite_parameter("s", 2, 3, num_supports=5)
    v = m.variable("v", t, lb=0, ub=100)
    pf = m.parameter_function("pf", np.sin, t)
    pf2 = m.parameter_function("pf2", lambda t, s: np.cos(t) * s, t, s)
    m.constraint(v + pf <= 100)
    m.constraint(v * 2 + pf * pf2 <= 100)
    data = transcribe.ExaMappingData()
    core = transcribe.exa_core(m, data)
    p1, p2 = data.param_mappings[pf], data.param_mappings[pf2]
    assert p1.length == 5 and p2.length == 25 and core.npar == 30
    tv, sv = np.array([0., .25, .5, .75, 1.]), np.array([2, 2.25, 2.5, 2.75, 3])
    np.testing.assert_array_equal(core.theta[p1.offset:p1.offset + 5], np.sin(tv))                      # :151
    np.testing.assert_array_equal(core.theta[p2.offset:p2.offset + 25],
                                  np.array([np.cos(a) * b for b in sv for a in tv]))                    # :164


def test_theta_literal_vector_of_solve_jl():
    """test/solve.jl:190-191 — the θ of pf2(t,s) = sin(t)·s + 0.2 on 3×3 supports, digit for digit."""
    m, (pf1, pf2) = cases.pfun()
    data = transcribe.ExaMappingData()
    core = transcribe.exa_core(m, data)
    p1, p2 = data.param_mappings[pf1], data.param_mappings[pf2]
    expected = [0.2, 1.158851077208406, 1.882941969615793, 0.2, 1.3985638465105075, 2.3036774620197416,
                0.2, 1.638276615812609, 2.7244129544236895]
    np.testing.assert_array_equal(core.theta[p1.offset:p1.offset + 3], np.sin([0.0, 0.5, 1.0]))
    np.testing.assert_array_equal(core.theta[p2.offset:p2.offset + 9], expected)


def test_domain_restriction_item_count():
    """test/transcription.jl:211-218."""
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, num_supports=10)
    y = m.variable("y", t)
    m.constraint(y ** 2 >= 2, restriction=DomainRestriction(lambda s: s >= 0.5, t))
    core = transcribe.exa_core(m)
    con = [tp for tp in core.templates if tp.kind == T_CON]
    assert len(con[0].items) == int((t.supports >= 0.5).sum())


def test_warmstart_problem_sizes_and_starts():
    """test/ipopt.jl:183-186: x0 == [10, 0, …] (length 51), y0 == zeros(70)."""
    core = transcribe.exa_core(cases.ode_5x5())
    expected = np.zeros(51)
    expected[0] = 10.0
    np.testing.assert_array_equal(core.x0, expected)
    assert core.ncon == 70


def test_objective_heuristics_accept_and_warn():
    """test/transcription.jl:177-209: nested measures with movable terms build one template;
    the 'not so good' forms take the expand_measures fallback with the reference's warning."""
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, num_supports=3)
    x1 = m.infinite_parameter("x1", -1, 1, num_supports=5)
    x2 = m.infinite_parameter("x2", -1, 1, num_supports=5)
    y = m.variable("y", t, x1, x2)
    q = m.variable("q", t)
    z = m.variable("z")
    x1_int = m.integral(y ** 2, x1)
    ok = [m.integral(m.integral(x1_int, x2), t),
          m.integral(m.integral(x1_int, x2) + 2 * q ** 2, t),
          m.integral(m.integral(x1_int, x2) * io.sin(q), t),
          m.integral(m.integral(x1_int, x2) + 2 * q, t),
          m.integral(m.integral(x1_int, x2) + io.sin(q), t)]
    for obj in ok:
        m.objective("min", obj)
        core = transcribe.exa_core(m)
        objs = [tp for tp in core.templates if tp.kind == T_OBJ]
        assert len(objs) == 1 and len(objs[0].items) == 3 * 5 * 5
    bad = [m.integral(m.integral(x1_int ** 2, x2), t), m.integral(m.integral(io.sin(x1_int), x2), t),
           m.integral(m.integral(x1_int * x1_int ** 2.3, x2), t),
           m.integral(m.integral(x1_int + m.integral(io.sin(y), x1), x2), t)]
    for obj in bad:
        m.objective("min", obj)
        with pytest.warns(UserWarning, match="Unable to convert objective measures"):
            core = transcribe.exa_core(m)                       # test/transcription.jl:205-208: still an ExaModel
        assert sum(tp.kind == T_OBJ for tp in core.templates) >= 1


def test_expand_measures_matches_explicit_quadrature(built):
    """Constrained measures and the objective fallback (transform.jl:433-435, 708-720) against
    a hand-written trapezoid sum."""
    from pyoracle import OracleModel
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, num_supports=3)
    x1 = m.infinite_parameter("x1", -1, 1, num_supports=4)
    y = m.variable("y", t, x1)
    q = m.variable("q", t)
    z = m.variable("z")
    m.objective("min", m.integral(io.sin(m.integral(y ** 2, x1)) + 0.3 * q, t) + z ** 2)
    with pytest.warns(UserWarning, match="Constrained measures"):
        m.constraint(m.integral(y, x1) + q <= 3.0)
        m.constraint(m.integral(m.integral(y * q, x1), t) == 1.0)
        core = transcribe.exa_core(m)
    om = OracleModel(core.to_blob())
    x = np.random.default_rng(0).normal(size=om.nvar)

    def trap(s):
        d = np.diff(s)
        c = np.zeros_like(s)
        c[:-1] += d / 2
        c[1:] += d / 2
        return c

    ct, cx = trap(np.linspace(0, 1, 3)), trap(np.linspace(-1, 1, 4))
    Y, Q, zz = x[1:13].reshape(4, 3).T, x[13:16], x[0]
    obj = sum(ct[i] * (np.sin(cx @ Y[i] ** 2) + 0.3 * Q[i]) for i in range(3)) + zz ** 2
    assert abs(om.obj(x) - obj) < 1e-12
    np.testing.assert_allclose(om.cons(x)[:3], [cx @ Y[i] + Q[i] for i in range(3)], rtol=1e-12, atol=1e-13)
    assert abs(om.cons(x)[3] - sum(ct[i] * (cx @ Y[i]) * Q[i] for i in range(3))) < 1e-13


def test_build_order_matches_build_exa_core():
    """transform.jl:777-794: user constraints → derivative approximations → objective."""
    core = cases.build_core("quadrotor_5")
    tags = [t.tag[0] for t in core.templates]
    assert tags == ["con"] * 18 + ["deriv"] * 9 + ["obj"]
    # product iterators run first-group-fastest (transform.jl:443-445)
    core = cases.build_core("pandemic_20x3")
    ode = core.templates[4]
    cols = ode.items.column("group_idx1"), ode.items.column("group_idx2")
    assert list(cols[0][:4]) == [1, 2, 3, 4] and list(cols[1][:4]) == [1, 1, 1, 1]


def test_orthogonal_collocation_structure_and_exactness(built):
    """ESCAPE34/quadrotor.jl:13-14,73 — OrthogonalCollocation(3) + constant_over_collocation.
    Structure follows the reference's own code (transform.jl:565-601: pairs (3,2),(5,4),…) and
    SURVEY Appendix B (2S−1 supports, 2(S−1) rows per derivative); the collocation equations
    themselves are an [EXT] restatement, so they are checked for what any order-n collocation
    must satisfy: exactness on polynomials of degree ≤ n − 1."""
    from infiniteexamodels.jl_amd import workloads
    from infiniteexamodels.jl_amd.infinite import OrthogonalCollocation
    from pyoracle import OracleModel
    S = 9
    core = transcribe.exa_core(workloads.quadrotor(S, collocation=3))
    assert core.nvar == 22 * (2 * S - 1)
    deriv = [t for t in core.templates if t.tag[0] == "deriv"]
    colloc = [t for t in core.templates if t.tag[0] == "colloc"]
    assert len(deriv) == 9 and all(len(t.items) == 2 * (S - 1) for t in deriv)
    assert len(colloc) == 4 and all(len(t.items) == S - 1 for t in colloc)
    assert [(r["i1"], r["i2"]) for r in colloc[0].items.records()[:3]] == [(3, 2), (5, 4), (7, 6)]
    for nodes, poly, dpoly in ((3, lambda t: 0.7 - 1.3 * t + 2.1 * t ** 2, lambda t: -1.3 + 4.2 * t),
                               (4, lambda t: 0.7 - 1.3 * t + 2.1 * t ** 2 - 0.4 * t ** 3,
                                lambda t: -1.3 + 4.2 * t - 1.2 * t ** 2)):
        m = InfiniteModel()
        tt = m.infinite_parameter("t", 0, 2, supports=[0, 0.3, 1.1, 2.0], derivative_method=OrthogonalCollocation(nodes))
        y = m.variable("y", tt)
        m.constraint(m.deriv(y, tt) == 0 * y)
        om = OracleModel(transcribe.exa_core(m).to_blob())
        ts = m.groups[0].supports[:, 0]
        assert len(ts) == 3 * (nodes - 1) + 1
        x = np.concatenate([poly(ts), dpoly(ts)])
        assert np.abs(om.cons(x)[len(ts):]).max() < 1e-13


def test_objective_sense_travels_as_metadata(built):
    """transform.jl:814-815: `minimize = objective_sense == MIN_SENSE` goes to ExaCore and from
    there to `meta.minimize` (NLPModels convention: obj/grad are NOT negated, the solver reads
    the flag)."""
    from pyoracle import OracleModel

    def make(sense):
        m = InfiniteModel()
        t = m.infinite_parameter("t", 0, 1, num_supports=4)
        y = m.variable("y", t, start=0.5)
        m.objective(sense, m.integral(y ** 2, t))
        return transcribe.exa_core(m)
    cmin, cmax = make("min"), make("max")
    assert cmin.minimize and not cmax.minimize
    omin, omax = OracleModel(cmin.to_blob()), OracleModel(cmax.to_blob())
    assert omin.minimize and not omax.minimize
    assert omin.obj(omin.x0) == omax.obj(omax.x0) > 0


@pytest.mark.parametrize("k", [0, 1, 2, 3, 4])
def test_problem_2_objective_forms_match_explicit_quadrature(k, built):
    """test/solve.jl:46-90: whatever the objective heuristics do with nested measures and finite
    terms (inner measure kept, product with a finite function, finite addend moved inside), the
    transcribed objective must equal the double trapezoid rule written out by hand."""
    from pyoracle import OracleModel
    core = cases.build_core(f"test_problem_2_obj{k}")
    om = OracleModel(core.to_blob())
    rng = np.random.default_rng(k)
    x = np.abs(om.x0 + rng.standard_normal(om.nvar)) + 0.1
    z = x[0]
    Y = x[1:26].reshape(5, 5)                 # Y[ix, it]: t runs fastest inside the slab
    tw = np.array([0.125, 0.25, 0.25, 0.25, 0.125])        # trapezoid on 5 uniform supports of [0, 1]
    xw = 2 * tw                                             # ... of [-1, 1]
    inner = (Y ** 2) @ tw                                   # I(x) = ∫ y² dt
    y01 = Y[4, 0]                                           # y(t = 0, x = 1)
    want = [xw @ (inner + 2 * z) + 2 * y01, xw @ (inner + 2 * z ** 2) + 2 * y01, xw @ (inner + np.sin(z ** 2)),
            xw @ (inner * np.cos(z)), xw @ (z * (inner + z ** 3))][k]
    assert abs(om.obj(x) - want) <= 1e-12 * max(1.0, abs(want))
