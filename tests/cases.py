"""Model cases shared by the CPU tests, the GPU parity tests and build() (which
pre-compiles their kernels for gfx950)."""
from __future__ import annotations

import numpy as np

from infiniteexamodels.jl_amd import infinite as io
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.infinite import DomainRestriction, InfiniteModel


def ode_5x5():
    """/root/reference/test/ipopt.jl:160-167 — nvar 51, ncon 70."""
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, num_supports=5)
    x = m.infinite_parameter("x", -1, 1, num_supports=5)
    y = m.variable("y", t, x, lb=0)
    z = m.variable("z", start=10)
    m.objective("min", m.integral(m.integral(y ** 2, t) + 2 * z, x))
    m.constraint(m.deriv(y, t) == io.sin(y) + z + 1.2)
    m.constraint(y + z <= 42 + t)
    return m


def test_problem_1():
    """/root/reference/test/solve.jl:2-14 (restricted constraint, point + semi-infinite vars)."""
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, num_supports=5)
    x = m.infinite_parameter("x", -1, 1, num_supports=5)
    y = m.variable("y", t, x, lb=0)
    z = m.variable("z", start=10)
    m.objective("min", m.integral(m.integral(y ** 2, t), x) + 2 * y(0, 1))
    m.constraint(m.deriv(y, t) == io.sin(y) + z + 1.2)
    m.constraint(y + z <= 42 + t, restriction=DomainRestriction(lambda s: (0 <= s) & (s <= 0.5), t))
    m.constraint(m.deriv(y(0, x), x) == 5)
    return m


def test_problem_1_oc3():
    """/root/reference/test/solve.jl:27-31 — the same model with OrthogonalCollocation(3) in t and
    an extra variable held constant over the collocation nodes."""
    m = test_problem_1()
    t = m.groups[0].prefs[0]
    m.set_derivative_method(t, io.OrthogonalCollocation(3))
    u = m.variable("u", t)
    m.constant_over_collocation(u, t)
    return m


def test_problem_2(objective: int = 0):
    """/root/reference/test/solve.jl:46-90 — five objective forms over the same constraints
    (0: the base objective, 1-4: the `objs` list exercising the measure heuristics)."""
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, num_supports=5)
    x = m.infinite_parameter("x", -1, 1, num_supports=5)
    y = m.variable("y", t, x, lb=0)
    z = m.variable("z", start=10)
    m.constraint(m.deriv(y, t) == io.sin(y) + z + 1.2)
    m.constraint(y + z <= 42 + t)
    m.constraint(m.deriv(y(0, x), x) == 5)
    inner = m.integral(y ** 2, t)
    objs = [lambda: m.integral(inner + 2 * z, x) + 2 * y(0, 1),
            lambda: m.integral(inner + 2 * z ** 2, x) + 2 * y(0, 1),
            lambda: m.integral(inner + io.sin(z ** 2), x),
            lambda: m.integral(inner * io.cos(z), x),
            lambda: m.integral(z * (inner + z ** 3), x)]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.objective("min", objs[objective]())
    return m


def pfun_full(ti=0.2):
    """/root/reference/test/solve.jl:93-122 — the whole "Parameter Function Problem": a piecewise
    parameter function of (t, s), a semi-infinite variable z(t, 2.5) and a measure of a parameter
    function inside a constraint (c5)."""
    def pf2f(t, s):
        return np.where(t <= 0.5, np.cos(t) * s - ti, np.sin(t) * s + ti)
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, num_supports=5)
    s = m.infinite_parameter("s", 2, 3, num_supports=5)
    v = m.variable("v", t, lb=0, ub=100)
    z = m.variable("z", t, s, lb=0, ub=100)
    pf = m.parameter_function("pf", np.sin, t)
    pf2 = m.parameter_function("pf2", pf2f, t, s)
    m.constraint(v + pf <= 100, name="c1")
    m.constraint(v * 2 + pf * pf2 <= 100, name="c2")
    m.constraint(v >= 0.2 * pf2, name="c3")
    m.constraint(z(t, 2.5) + pf2 * pf <= 40, name="c4")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.constraint(v * m.integral(pf2, s) <= 100, name="c5")
    m.objective("min", m.integral(v * pf, t) + m.integral(m.integral(0.5 * z * pf2, t), s))
    return m


def rosenbrock(p1=100.0, p2=1.0):
    """/root/reference/test/solve.jl:134-143 — finite parameters p1, p2."""
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, num_supports=3)
    P1 = m.finite_parameter("p1", p1)
    P2 = m.finite_parameter("p2", p2)
    x = [m.variable(f"x[{i + 1}]", t) for i in range(2)]
    m.objective("min", P1 * m.integral((x[1] - x[0] ** 2) ** 2, t) + m.integral((P2 - x[0]) ** 2, t))
    for i, ub in enumerate([0.5, 3.0]):
        m.constraint(x[i] <= ub)
    m.constraint(x[0] * x[1] >= 1.0)
    m.constraint(x[0] + x[1] ** 2 >= 0.0)
    return m, (P1, P2)


def pfun(k=0.2, f1=np.sin):
    """/root/reference/test/solve.jl:173-186 — parameter functions pf1(t), pf2(t, s)."""
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, num_supports=3)
    s = m.infinite_parameter("s", 2, 3, num_supports=3)
    v = m.variable("v", t, lb=0, ub=100)
    z = m.variable("z", t, s, lb=0, ub=100)
    pf1 = m.parameter_function("pf1", f1, t)
    pf2 = m.parameter_function("pf2", lambda t, s: np.sin(t) * s + k, t, s)
    m.constraint(v + pf1 <= 100, name="c1")
    m.constraint(v * 2 + pf1 * pf2 <= 100, name="c2")
    m.constraint(v >= 0.5 * pf2, name="c3")
    m.constraint(z(t, 2.5) + pf2 * pf1 <= 40, name="c4")
    m.objective("min", m.integral(v * pf1, t) + m.integral(m.integral(0.5 * z * pf2, t), s))
    return m, (pf1, pf2)


def operator_zoo():
    """Every operator of /root/reference/src/operators.jl:3-44 in one model."""
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, num_supports=70)
    a = m.variable("a", t, start=0.6)
    b = m.variable("b", t, start=1.7)
    names = ["inv", "sqrt", "cbrt", "abs", "abs2", "exp", "exp2", "log", "log2", "log10", "log1p",
             "sin", "cos", "tan", "asin", "acos", "csc", "sec", "cot", "atan", "acot",
             "sind", "cosd", "tand", "cscd", "secd", "cotd", "atand", "acotd",
             "sinh", "cosh", "tanh", "csch", "sech", "coth", "atanh"]
    for n in names:
        f = getattr(io, n)
        m.constraint(f(a) * b + f(a * 0.5) >= -1e3)
    m.constraint(io.acoth(b) * a >= -1e3)
    m.constraint(a ** b + b ** 2.5 + 2.0 ** a + a / b + (a * b) ** 3 >= -1e3)
    m.constraint(-io.sin(a) + (+b) - a / 2.0 + 3.0 / b >= -1e3)
    m.objective("min", m.integral(io.exp(a) * io.sin(b) + (a - b) ** 2 / (1 + b ** 2), t))
    return m


def irregular():
    """Explicit-column iterators: a domain restriction that is not a contiguous range."""
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, num_supports=130)
    y = m.variable("y", t, start=1.0)
    w = m.variable("w", start=0.5)
    m.constraint(y ** 2 * w >= 2, restriction=DomainRestriction(lambda s: np.sin(40 * s) >= 0.2, t))
    m.constraint(m.deriv(y, t) == -y * w)
    m.objective("min", m.integral(y ** 2, t) + w ** 2)
    return m


def small_cases():
    """name -> (core builder, positive?)"""
    return {
        "quadrotor_1": lambda: workloads.quadrotor(1),      # degenerate: the nine difference templates have NO items
        "quadrotor_5": lambda: workloads.quadrotor(5),
        "quadrotor_100": lambda: workloads.quadrotor(100),
        "quadrotor_1000": lambda: workloads.quadrotor(1000),
        "quadrotor_oc3_40": lambda: workloads.quadrotor(40, collocation=3),
        "quadrotor_oc3_700": lambda: workloads.quadrotor(700, collocation=3),
        "pandemic_20x3": lambda: workloads.pandemic(20, 3),
        "pandemic_300x7": lambda: workloads.pandemic(300, 7),
        "pandemic_oc3_20x3": lambda: workloads.pandemic(20, 3, collocation=3),    # examples/pandemic.jl: collocation x scenarios (node x element x scenario boxes)
        "pandemic_oc3_40x3": lambda: workloads.pandemic(40, 3, collocation=3),    # ... with a time axis of more than 64 supports: a 2-D grid the boxes are folded onto
        "farmer_1": lambda: workloads.farmer(1),
        "farmer_5": lambda: workloads.farmer(5),
        "farmer_1000": lambda: workloads.farmer(1000),
        "opf_7": lambda: workloads.opf(7),
        "opf_600": lambda: workloads.opf(600),
        # the remaining models of /root/reference/examples: interior waypoints (point variables at
        # supports added for them), a 3-dimensional dependent parameter with a MAX expectation
        # objective, collocation(4) on a non-uniform grid with a point-variable objective
        "hovercraft": lambda: workloads.hovercraft(),
        "hovercraft_oc4": lambda: workloads.hovercraft(21, collocation=4),
        "three_node_50": lambda: workloads.three_node_design(50),
        "kinetic_20": lambda: workloads.kinetic_control(20),
        "ode_5x5": ode_5x5,
        "test_problem_1": test_problem_1,
        "test_problem_1_oc3": test_problem_1_oc3,
        "test_problem_2_obj0": lambda: test_problem_2(0),
        "test_problem_2_obj1": lambda: test_problem_2(1),
        "test_problem_2_obj2": lambda: test_problem_2(2),
        "test_problem_2_obj3": lambda: test_problem_2(3),
        "test_problem_2_obj4": lambda: test_problem_2(4),
        "pfun_full": pfun_full,
        "rosenbrock": lambda: rosenbrock()[0],
        "pfun": lambda: pfun()[0],
        "operator_zoo": operator_zoo,
        "irregular": irregular,
        "wide_rows": None,
        "many_templates": None,
    }


def eval_point_for(name, om, seed=0):
    """Seeded evaluation point kept inside every operator's domain."""
    rng = np.random.default_rng(seed)
    x = om.x0 + 0.1 * rng.standard_normal(om.nvar)
    if name.startswith(("quadrotor", "opf", "hovercraft")):
        pass
    elif name.startswith("kinetic"):
        x = om.x0 + 0.01 * rng.standard_normal(om.nvar) * np.maximum(1.0, np.abs(om.x0))   # T stays near 333 K
    elif name == "operator_zoo":
        n = om.nvar // 2
        x = np.concatenate([0.3 + 0.5 * rng.random(n), 1.3 + 0.6 * rng.random(om.nvar - n)])
    else:
        x = np.abs(x) + 0.05
    y = np.random.default_rng(seed + 1).standard_normal(om.ncon)
    return x, y


def pandemic_two_controls(num_supports, num_scenarios):
    """The SIR model of ESCAPE34/pandemic.jl with a SECOND control that lives on t alone (a treatment rate v(t) moving infected
    to recovered): two hubs per time block for the chain KKT solver's hub border — not a reference model, a shape test."""
    gamma, beta, N = 0.303, 0.727, 1e5
    im = InfiniteModel()
    t = im.infinite_parameter("t", 0.0, 200.0, num_supports=num_supports)
    xi = im.infinite_parameter("ξ", supports=np.linspace(0.1, 0.6, num_scenarios))
    im.add_supports(t, [0.001, 0.002, 0.004, 0.008, 0.02, 0.04, 0.08, 0.2, 0.4, 0.8])
    s, e, i, r = (im.variable(n, t, xi, lb=0) for n in "seir")
    u = im.variable("u", t, lb=0, ub=0.8, start=0.2)
    v = im.variable("v", t, lb=0, ub=0.3, start=0.1)
    im.objective("min", im.integral(u + 0.5 * v * v, t))
    im.constraint(s(0, xi) == 1 - 1 / N); im.constraint(e(0, xi) == 1 / N); im.constraint(i(0, xi) == 0); im.constraint(r(0, xi) == 0)
    d = lambda w: im.deriv(w, t)
    im.constraint(d(s) == -(1 - u) * beta * s * i)
    im.constraint(d(e) == (1 - u) * beta * s * i - xi * e)
    im.constraint(d(i) == xi * e - gamma * i - v * i)
    im.constraint(d(r) == gamma * i + v * i)
    im.constraint(i <= 0.02)
    return im


def extra_cases():
    """Models only some test modules ask for by name (not part of the every-entry-point sweep over small_cases())."""
    return {
        # a 2-D support grid whose time blocks are too large for one chain (17 N_xi + 1 unknowns): the chain KKT solver runs one
        # chain per scenario (lanes) with u(t) in the border — the shape of the reference's own ladder, ESCAPE34/run_cases_gpu.jl:99-102
        "pandemic_100x7": lambda: workloads.pandemic(90, 7),
        # ... and with more time supports than a dense border holds (128): u(t) as span-sparse hubs (BASELINE config 3's shape)
        "pandemic_200x24": lambda: workloads.pandemic(190, 24),
        "pandemic2_150x8": lambda: pandemic_two_controls(140, 8),      # two hubs per time block (u(t), v(t)): 300 border unknowns
        "pandemic2_20x3": lambda: pandemic_two_controls(10, 3),
    }


def build_core(name):
    if name == "wide_rows":
        return wide_rows()
    if name == "many_templates":
        return many_templates()
    if name in extra_cases():
        return transcribe.exa_core(extra_cases()[name]())
    return transcribe.exa_core(small_cases()[name]())


def wide_rows():
    """One constraint with 14 variables multiplied pairwise-nonlinearly: > 100 Hessian slots per
    item, more than the LDS staging budget — exercises the direct-store fallback."""
    from infiniteexamodels.jl_amd import DataSource, ExaCore, Items, FUNCS
    core = ExaCore()
    n = 300
    vs = [core.add_var(n, start=0.1 * (k + 1)) for k in range(14)]
    ds = DataSource()
    it = Items.from_supports("i", n, {"t": np.linspace(0, 1, n)}, group_id=1)
    prod = vs[0][ds.i]
    for v in vs[1:]:
        prod = prod * FUNCS["cos"](v[ds.i]) + v[ds.i]
    core.add_con(FUNCS["sin"](prod), it)
    core.add_con(vs[0][ds.i] * vs[1][ds.i], it)
    core.add_obj(FUNCS["abs2"](vs[2][ds.i]), it)
    return core


def many_templates():
    """200 distinct constraint templates over one grid: the kernel's offset tables no longer fit
    the 4 KB argument block and travel through device memory instead."""
    from infiniteexamodels.jl_amd import DataSource, ExaCore, Items, FUNCS
    core = ExaCore()
    n = 260
    vs = [core.add_var(n, start=0.05 * (k + 1)) for k in range(40)]
    ds = DataSource()
    it = Items.from_supports("i", n, {"t": np.linspace(0, 1, n)}, group_id=1)
    for k in range(200):
        a, b, c = vs[k % 40], vs[(3 * k + 1) % 40], vs[(7 * k + 2) % 40]
        core.add_con(a[ds.i] * FUNCS["sin"](b[ds.i]) + (0.1 + 0.01 * k) * c[ds.i] * c[ds.i], it)
    core.add_obj(FUNCS["abs2"](vs[0][ds.i]) + vs[1][ds.i] * vs[2][ds.i], it)
    return core
