"""The reference's benchmark models, stated with the modelling layer.

Each function restates one file of ``/root/reference/examples`` / ``ESCAPE34`` line by
line (citations inline); sizes are arguments so the same statement serves the
plumbing case (100 supports) and the device cases (10⁵–10⁶ supports).
"""
from __future__ import annotations

import numpy as np

from . import infinite as io
from .infinite import InfiniteModel


def quadrotor(num_supports: int = 100, backend=None, supports=None, collocation: int = 0) -> InfiniteModel:
    """``/root/reference/examples/quadrotor.jl:6-77`` — 9 states, 4 controls, T = 60,
    backward finite differences (InfiniteOpt default).  ``collocation = 3`` gives the
    ``ESCAPE34/quadrotor.jl:13-14,73`` variant: ``OrthogonalCollocation(3)`` and controls held
    constant over each element (``constant_over_collocation.(u, t)``)."""
    n, p, T = 9, 4, 60.0
    im = InfiniteModel(backend)
    method = io.OrthogonalCollocation(collocation) if collocation else None
    if supports is not None:   # explicit support window (shard.py)
        t = im.infinite_parameter("t", 0.0, T, supports=supports, derivative_method=method)
    else:
        t = im.infinite_parameter("t", 0.0, T, num_supports=num_supports, derivative_method=method)   # :19
    d1 = im.parameter_function("d1", lambda t: np.sin(2 * np.pi * t / T), t)        # :21
    d3 = im.parameter_function("d3", lambda t: 2 * np.sin(4 * np.pi * t / T), t)    # :22
    d5 = im.parameter_function("d5", lambda t: 2 * (t / T), t)                      # :23
    x = [im.variable(f"x[{i + 1}]", t) for i in range(n)]                           # :29
    u = [im.variable(f"u[{i + 1}]", t, start=0) for i in range(p)]                  # :31
    X = [None] + x   # 1-based, as in the reference
    U = [None] + u
    cos, sin, tan = io.cos, io.sin, io.tan
    im.objective("min", im.integral(                                               # :34-40
        (X[1] - d1) ** 2 + (X[3] - d3) ** 2 + (X[5] - d5) ** 2 + X[7] ** 2 + X[8] ** 2 + X[9] ** 2
        + 0.1 * (U[1] ** 2 + U[2] ** 2 + U[3] ** 2 + U[4] ** 2), t))
    for i in range(1, n + 1):                                                       # :41
        im.constraint(X[i](0) == 0)
    d = lambda v: im.deriv(v, t)
    im.constraint(d(X[1]) == X[2])                                                  # :42-45
    im.constraint(d(X[2]) == U[1] * cos(X[7]) * sin(X[8]) * cos(X[9]) + U[1] * sin(X[7]) * sin(X[9]))
    im.constraint(d(X[3]) == X[4])
    im.constraint(d(X[4]) == U[1] * cos(X[7]) * sin(X[8]) * sin(X[9]) - U[1] * sin(X[7]) * cos(X[9]))
    im.constraint(d(X[5]) == X[6])
    im.constraint(d(X[6]) == U[1] * cos(X[7]) * cos(X[8]) - 9.8)
    im.constraint(d(X[7]) == U[2] * cos(X[7]) / cos(X[8]) + U[3] * sin(X[7]) / cos(X[8]))
    im.constraint(d(X[8]) == -U[2] * sin(X[7]) + U[3] * cos(X[7]))
    im.constraint(d(X[9]) == U[2] * cos(X[7]) * tan(X[8]) + U[3] * sin(X[7]) * tan(X[8]) + U[4])  # :74-77
    if collocation:
        for uk in u:                                                                # ESCAPE34/quadrotor.jl:73
            im.constant_over_collocation(uk, t)
    return im


def pandemic(num_supports: int = 100, num_scenarios: int = 4, backend=None, xi_supports=None, collocation: int = 0) -> InfiniteModel:
    """``/root/reference/ESCAPE34/pandemic.jl:4-34`` — SIR optimal control.
    ξ supports are synthetic (equispaced in [0.1, 0.6]) instead of Julia-RNG
    ``Uniform(0.1, 0.6)`` draws (``:16``).  ``collocation = 3`` gives ``/root/reference/examples/pandemic.jl:11,17,33``:
    ``OrthogonalCollocation(3)`` on the non-uniform t grid and ``constant_over_collocation(u, t)`` — derivative rows over
    node × element × scenario boxes."""
    from . import infinite as io
    gamma, beta, N = 0.303, 0.727, 1e5                                              # :8-10
    extra_ts = [0.001, 0.002, 0.004, 0.008, 0.02, 0.04, 0.08, 0.2, 0.4, 0.8]         # :11
    im = InfiniteModel(backend)
    t = im.infinite_parameter("t", 0.0, 200.0, num_supports=num_supports,
                              **({"derivative_method": io.OrthogonalCollocation(collocation)} if collocation else {}))   # :15
    xi = im.infinite_parameter("ξ", supports=(xi_supports if xi_supports is not None
                                              else np.linspace(0.1, 0.6, num_scenarios)))   # :16
    im.add_supports(t, extra_ts)                                                     # :17
    s = im.variable("s", t, xi, lb=0)                                                # :18-21
    e = im.variable("e", t, xi, lb=0)
    i = im.variable("i", t, xi, lb=0)
    r = im.variable("r", t, xi, lb=0)
    u = im.variable("u", t, lb=0, ub=0.8, start=0.2)                                 # :22
    im.objective("min", im.integral(u, t))                                           # :23
    im.constraint(s(0, xi) == 1 - 1 / N)                                             # :24-27
    im.constraint(e(0, xi) == 1 / N)
    im.constraint(i(0, xi) == 0)
    im.constraint(r(0, xi) == 0)
    d = lambda v: im.deriv(v, t)
    im.constraint(d(s) == -(1 - u) * beta * s * i, name="s_constr")                  # :28
    im.constraint(d(e) == (1 - u) * beta * s * i - xi * e, name="e_constr")          # :29
    im.constraint(d(i) == xi * e - gamma * i, name="i_constr")                       # :30
    im.constraint(d(r) == gamma * i, name="r_constr")                                # :31
    im.constraint(i <= 0.02, name="imax_constr")                                     # :32
    if collocation:
        im.constant_over_collocation(u, t)                                           # examples/pandemic.jl:33
    return im


def farmer_supports(num_scenarios: int, seed: int = 42) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return np.column_stack([rng.uniform(0, 5, num_scenarios), rng.uniform(0, 5, num_scenarios),
                            rng.uniform(10, 30, num_scenarios)])


def farmer(num_scenarios: int = 1000, seed: int = 42, backend=None, supports=None) -> InfiniteModel:
    """``/root/reference/examples/2stage_example.jl:7-37`` — two-stage stochastic farmer.
    ξ supports: seeded numpy uniforms on the ranges of ``Ξ`` (``:15``)."""
    alpha = [150, 230, 260]
    beta = [238, 210, 0]
    lam = [170, 150, 36]
    dem = [200, 240, 0]
    xbar, wbar3, ybar3 = 500, 6000, 0
    supp = supports if supports is not None else farmer_supports(num_scenarios, seed)
    im = InfiniteModel(backend)
    xi = im.dependent_parameters(["ξ[1]", "ξ[2]", "ξ[3]"], supp)                      # :21
    x = [im.variable(f"x[{c + 1}]", lb=0, ub=xbar) for c in range(3)]                # :23
    y = [im.variable(f"y[{c + 1}]", *xi, lb=0) for c in range(3)]                    # :25
    w = [im.variable(f"w[{c + 1}]", *xi, lb=0) for c in range(3)]                    # :26
    second = sum(beta[c] * y[c] for c in range(3)) - sum(lam[c] * w[c] for c in range(3))
    im.objective("min", sum(alpha[c] * x[c] for c in range(3)) + im.expect(second, xi))  # :28
    im.constraint(x[0] + x[1] + x[2] <= xbar)                                        # :31
    for c in range(3):                                                               # :33
        im.constraint(xi[c] * x[c] + y[c] - w[c] >= dem[c])
    im.constraint(w[2] <= wbar3)                                                     # :35
    im.constraint(y[2] <= ybar3)                                                     # :36
    return im


# ---------------------------------------------------------------------------
# stochastic AC-OPF (ESCAPE34/opf.jl)
# ---------------------------------------------------------------------------
# The reference downloads pglib_opf_case3_lmbd.m at run time (ESCAPE34/opf.jl:15-18) and reads
# it through PowerModels; neither is available here, so a 3-bus / 3-generator / 3-branch network
# of the same shape is embedded in per-unit (baseMVA 100).  Values are synthetic stand-ins in
# the range of that case, NOT a copy of the pglib file.
OPF_NETWORK = dict(
    bus={1: dict(vmin=0.9, vmax=1.1, pd=1.10, qd=0.40, gs=0.0, bs=0.0),
         2: dict(vmin=0.9, vmax=1.1, pd=1.10, qd=0.40, gs=0.0, bs=0.0),
         3: dict(vmin=0.9, vmax=1.1, pd=0.95, qd=0.50, gs=0.0, bs=0.0)},
    ref_buses=[1],
    gen={1: dict(bus=1, pmin=0.0, pmax=20.0, qmin=-10.0, qmax=10.0, cost=(1100.0, 500.0, 0.0)),
         2: dict(bus=2, pmin=0.0, pmax=20.0, qmin=-10.0, qmax=10.0, cost=(850.0, 120.0, 0.0)),
         3: dict(bus=3, pmin=0.0, pmax=0.0, qmin=-10.0, qmax=10.0, cost=(0.0, 0.0, 0.0))},
    branch={1: dict(f_bus=1, t_bus=3, r=0.065, x=0.620, b_c=0.450, rate_a=90.0, angmin=-0.5236, angmax=0.5236),
            2: dict(f_bus=3, t_bus=2, r=0.025, x=0.750, b_c=0.700, rate_a=0.5, angmin=-0.5236, angmax=0.5236),
            3: dict(f_bus=1, t_bus=2, r=0.042, x=0.900, b_c=0.300, rate_a=90.0, angmin=-0.5236, angmax=0.5236)},
)


def opf_supports(num_supports: int, seed: int = 0, net=OPF_NETWORK) -> np.ndarray:
    """Load perturbations θ ~ N(0, (0.1·[Pd; Qd])²) (opf.jl:46-52, :112), seeded numpy draws."""
    bus = net["bus"]
    std = 0.1 * np.array([bus[i]["pd"] for i in sorted(bus)] + [bus[i]["qd"] for i in sorted(bus)])
    return np.random.default_rng(seed).standard_normal((num_supports, len(std))) * std


def opf(num_supports: int = 100, seed: int = 0, backend=None, supports=None, net=OPF_NETWORK) -> InfiniteModel:
    """``/root/reference/ESCAPE34/opf.jl:36-285`` — two-stage stochastic AC-OPF: first-stage
    (finite) dispatch, per-scenario recourse network coupled through ramping rows.  Julia ``Dict``
    iteration order of buses/generators is replaced by sorted keys."""
    bus, gen, brs = net["bus"], net["gen"], net["branch"]
    nbus = len(bus)
    branch = []
    for l in sorted(brs):
        br = brs[l]
        den = br["r"] ** 2 + br["x"] ** 2
        g, b = br["r"] / den, -br["x"] / den              # PowerModels.calc_branch_y
        tr, ti = 1.0, 0.0                                  # calc_branch_t (tap 1, shift 0)
        branch.append(dict(f_bus=br["f_bus"], t_bus=br["t_bus"], f_idx=(l, br["f_bus"], br["t_bus"]),
                           t_idx=(l, br["t_bus"], br["f_bus"]), g=g, b=b, tr=tr, ti=ti, ttm=tr ** 2 + ti ** 2,
                           g_fr=0.0, b_fr=br["b_c"] / 2, g_to=0.0, b_to=br["b_c"] / 2,
                           angmin=br["angmin"], angmax=br["angmax"], rate_a=br["rate_a"]))
    arcs = [b["f_idx"] for b in branch] + [b["t_idx"] for b in branch]
    bus_arcs = {i: [a for a in arcs if a[1] == i] for i in bus}
    bus_gens = {i: [k for k in sorted(gen) if gen[k]["bus"] == i] for i in bus}
    cos, sin = io.cos, io.sin

    im = InfiniteModel(backend)
    # first stage variables (opf.jl:83-109)
    va0 = {i: im.variable(f"va0[{i}]") for i in sorted(bus)}
    vm0 = {i: im.variable(f"vm0[{i}]", lb=bus[i]["vmin"], ub=bus[i]["vmax"], start=1.0) for i in sorted(bus)}
    pg0 = {k: im.variable(f"pg0[{k}]", lb=gen[k]["pmin"], ub=gen[k]["pmax"]) for k in sorted(gen)}
    qg0 = {k: im.variable(f"qg0[{k}]", lb=gen[k]["qmin"], ub=gen[k]["qmax"]) for k in sorted(gen)}
    p0 = {a: im.variable(f"p0{a}", lb=-brs[a[0]]["rate_a"], ub=brs[a[0]]["rate_a"]) for a in arcs}
    q0 = {a: im.variable(f"q0{a}", lb=-brs[a[0]]["rate_a"], ub=brs[a[0]]["rate_a"]) for a in arcs}
    # second stage (opf.jl:112-140)
    supp = supports if supports is not None else opf_supports(num_supports, seed, net)
    th = im.dependent_parameters([f"θ[{i + 1}]" for i in range(2 * nbus)], supp)
    va = {i: im.variable(f"va[{i}]", *th) for i in sorted(bus)}
    vm = {i: im.variable(f"vm[{i}]", *th, lb=bus[i]["vmin"], ub=bus[i]["vmax"], start=1.0) for i in sorted(bus)}
    pg = {k: im.variable(f"pg[{k}]", *th, lb=gen[k]["pmin"], ub=gen[k]["pmax"]) for k in sorted(gen)}
    qg = {k: im.variable(f"qg[{k}]", *th, lb=gen[k]["qmin"], ub=gen[k]["qmax"]) for k in sorted(gen)}
    p = {a: im.variable(f"p{a}", *th, lb=-brs[a[0]]["rate_a"], ub=brs[a[0]]["rate_a"]) for a in arcs}
    q = {a: im.variable(f"q{a}", *th, lb=-brs[a[0]]["rate_a"], ub=brs[a[0]]["rate_a"]) for a in arcs}

    im.objective("min", sum(gen[k]["cost"][0] * pg0[k] ** 2 + gen[k]["cost"][1] * pg0[k] + gen[k]["cost"][2]
                            for k in sorted(gen)))                                  # :142-146

    def network(VA, VM, PG, QG, P, Q, load_p, load_q):
        for i in net["ref_buses"]:
            im.constraint(VA[i] == 0)                                               # :149 / :213
        for br in branch:                                                           # :150-157
            f, t = br["f_bus"], br["t_bus"]
            im.constraint(P[br["f_idx"]] ==
                          (br["g"] + br["g_fr"]) / br["ttm"] * VM[f] ** 2 +
                          (-br["g"] * br["tr"] + br["b"] * br["ti"]) / br["ttm"] * (VM[f] * VM[t] * cos(VA[f] - VA[t])) +
                          (-br["b"] * br["tr"] - br["g"] * br["ti"]) / br["ttm"] * (VM[f] * VM[t] * sin(VA[f] - VA[t])))
        for br in branch:                                                           # :158-165
            f, t = br["f_bus"], br["t_bus"]
            im.constraint(Q[br["f_idx"]] ==
                          -(br["b"] + br["b_fr"]) / br["ttm"] * VM[f] ** 2 -
                          (-br["b"] * br["tr"] - br["g"] * br["ti"]) / br["ttm"] * (VM[f] * VM[t] * cos(VA[f] - VA[t])) +
                          (-br["g"] * br["tr"] + br["b"] * br["ti"]) / br["ttm"] * (VM[f] * VM[t] * sin(VA[f] - VA[t])))
        for br in branch:                                                           # :168-175
            f, t = br["f_bus"], br["t_bus"]
            im.constraint(P[br["t_idx"]] ==
                          (br["g"] + br["g_to"]) * VM[t] ** 2 +
                          (-br["g"] * br["tr"] - br["b"] * br["ti"]) / br["ttm"] * (VM[t] * VM[f] * cos(VA[t] - VA[f])) +
                          (-br["b"] * br["tr"] + br["g"] * br["ti"]) / br["ttm"] * (VM[t] * VM[f] * sin(VA[t] - VA[f])))
        for br in branch:                                                           # :176-183
            f, t = br["f_bus"], br["t_bus"]
            im.constraint(Q[br["t_idx"]] ==
                          -(br["b"] + br["b_to"]) * VM[t] ** 2 -
                          (-br["b"] * br["tr"] + br["g"] * br["ti"]) / br["ttm"] * (VM[t] * VM[f] * cos(VA[t] - VA[f])) +
                          (-br["g"] * br["tr"] - br["b"] * br["ti"]) / br["ttm"] * (VM[t] * VM[f] * sin(VA[t] - VA[f])))
        for br in branch:                                                           # :186 / :250
            im.constraint_interval(VA[br["f_bus"]] - VA[br["t_bus"]], br["angmin"], br["angmax"])
        for br in branch:                                                           # :189-190
            im.constraint(P[br["f_idx"]] ** 2 + Q[br["f_idx"]] ** 2 <= br["rate_a"])
        for br in branch:
            im.constraint(P[br["t_idx"]] ** 2 + Q[br["t_idx"]] ** 2 <= br["rate_a"])
        for i in sorted(bus):                                                       # :193-210 / :257-278
            im.constraint(sum(P[a] for a in bus_arcs[i]) ==
                          load_p(i) + sum(PG[g] for g in bus_gens[i]) - bus[i]["pd"] - bus[i]["gs"] * VM[i] ** 2)
            im.constraint(sum(Q[a] for a in bus_arcs[i]) ==
                          load_q(i) + sum(QG[g] for g in bus_gens[i]) - bus[i]["qd"] + bus[i]["bs"] * VM[i] ** 2)

    network(va0, vm0, pg0, qg0, p0, q0, lambda i: 0.0, lambda i: 0.0)
    network(va, vm, pg, qg, p, q, lambda i: th[i - 1], lambda i: th[nbus + i - 1])
    for k in sorted(gen):                                                           # :282-283
        d = 0.1 * (gen[k]["pmax"] - gen[k]["pmin"])
        im.constraint_interval(pg0[k] - pg[k], -d, d)
    for k in sorted(gen):
        d = 0.1 * (gen[k]["qmax"] - gen[k]["qmin"])
        im.constraint_interval(qg0[k] - qg[k], -d, d)
    return im


def hovercraft(num_supports: int = 101, collocation: int = 0, backend=None) -> InfiniteModel:
    """``/root/reference/examples/hovercraft_example.jl:6-28`` — waypoint tracking: point variables at
    interior times (25, 50 are not on the 101-point grid, so they become supports of their own) and
    at the horizon's ends.  The example's last assignment selects backward finite differences;
    ``collocation = 4`` gives its first (overwritten) choice, ``OrthogonalCollocation(4)`` with
    ``constant_over_collocation.(u, t)``."""
    xw = np.array([[1.0, 4.0, 6.0, 1.0], [1.0, 3.0, 0.0, 1.0]])
    tw = [0.0, 25.0, 50.0, 60.0]
    im = InfiniteModel(backend)
    method = io.OrthogonalCollocation(collocation) if collocation else None
    t = im.infinite_parameter("t", 0.0, 60.0, num_supports=num_supports, derivative_method=method)
    im.add_supports(t, [v for v in tw if v not in (0.0, 60.0)])      # x[i](tw[j]) creates the supports 25, 50
    x = [im.variable(f"x[{i + 1}]", t) for i in range(2)]
    v = [im.variable(f"v[{i + 1}]", t) for i in range(2)]
    u = [im.variable(f"u[{i + 1}]", t, start=0) for i in range(2)]
    im.objective("min", im.integral(u[0] ** 2 + u[1] ** 2, t))
    for i in range(2):
        im.constraint(v[i](0) == 0)
    for i in range(2):
        im.constraint(im.deriv(x[i], t) == v[i])
    for i in range(2):
        im.constraint(im.deriv(v[i], t) == u[i])
    for i in range(2):
        for j, tj in enumerate(tw):
            im.constraint(x[i](tj) == xw[i, j])
    if collocation:
        for ui in u:
            im.constant_over_collocation(ui, t)
    return im


def three_node_design(num_supports: int = 1000, seed: int = 42, backend=None) -> InfiniteModel:
    """``/root/reference/examples/3node_design.jl:6-31`` — stochastic design over a 3-dimensional
    DEPENDENT parameter θ ~ MvNormal (synthetic draws: numpy's generator, not Julia's), an
    expectation objective with MAX sense, finite design variables in infinite constraints."""
    th_nom = np.array([0.0, 60.0, 10.0])
    covar = np.diag([80.0, 80.0, 120.0])
    c = np.ones(3) / np.sqrt(3.0)
    c_max, U = 5.0, 10000.0
    supp = np.random.default_rng(seed).multivariate_normal(th_nom, covar, size=num_supports)
    im = InfiniteModel(backend)
    th = im.dependent_parameters([f"θ[{i + 1}]" for i in range(3)], supp)
    y = im.variable("y", *th, lb=0, ub=1)
    z = [im.variable(f"z[{i + 1}]", *th) for i in range(3)]
    d = [im.variable(f"d[{i + 1}]", lb=0) for i in range(3)]
    im.objective("max", im.expect(1 - y, th))
    im.constraint(-z[0] - 35 - d[0] <= y * U)
    im.constraint(z[0] - 35 - d[0] <= y * U)
    im.constraint(-z[1] - 50 - d[1] <= y * U)
    im.constraint(z[0] - 50 - d[1] <= y * U)
    im.constraint(-z[2] <= y * U)
    im.constraint(z[2] - 100 - d[2] <= y * U)
    im.constraint(z[0] - th[0] == 0)
    im.constraint(-z[0] - z[1] + z[2] - th[1] == 0)
    im.constraint(z[1] - th[2] == 0)
    im.constraint(c[0] * d[0] + c[1] * d[1] + c[2] * d[2] <= c_max)
    return im


def kinetic_control(num_supports: int = 100, backend=None) -> InfiniteModel:
    """``/root/reference/examples/kinetic_control.jl:6-31`` — batch-reactor temperature control:
    ``OrthogonalCollocation(4)`` on a NON-uniform grid (seven extra supports near t = 0), a point
    variable as the MAX objective, Arrhenius rates (exp of a quotient), ``constant_over_collocation``."""
    A = [3.6362e6, 2.5212e16, 190.6879, 8.7409e24]
    Ea = [10000.0, 25000.0, 5000.0, 40000.0]
    R = 1.987
    T_lower, T_upper = 273.0 + 40, 273.0 + 60
    c0 = [1.0, 0.0, 0.0]
    Tr = [273.0 + v for v in (30, 40, 50, 70)]
    kr = [A[j] * np.exp(-Ea[j] / R / Tr[j]) for j in range(4)]
    tf = 3.0
    im = InfiniteModel(backend)
    t = im.infinite_parameter("t", 0.0, tf, num_supports=num_supports, derivative_method=io.OrthogonalCollocation(4))
    im.add_supports(t, [0.00001, 0.00005, 0.0001, 0.0005, 0.001, 0.01, 0.1])
    c = [im.variable(f"c[{i + 1}]", t, lb=0, ub=1, start=c0[i]) for i in range(3)]
    T = im.variable("T", t, lb=T_lower, ub=T_upper, start=T_upper)
    im.objective("max", c[1](tf))
    for i in range(3):
        im.constraint(c[i](0) == c0[i])
    k = [kr[j] * io.exp(Ea[j] / R * (1 / Tr[j] - 1 / T)) for j in range(4)]
    r1 = c[0] * k[0] - c[1] * k[1]
    r2 = c[0] * k[2] - c[2] * k[3]
    im.constraint(im.deriv(c[0], t) == -r1 - r2)
    im.constraint(im.deriv(c[1], t) == r1)
    im.constraint(im.deriv(c[2], t) == r2)
    im.constant_over_collocation(T, t)
    return im
