"""The reference's benchmark models, stated with the modelling layer.

Each function restates one file of ``/root/reference/examples`` / ``ESCAPE34`` line by
line (citations inline); sizes are arguments so the same statement serves the
plumbing case (100 supports) and the device cases (10⁵–10⁶ supports).
"""
from __future__ import annotations

import numpy as np

from . import infinite as io
from .infinite import InfiniteModel


def quadrotor(num_supports: int = 100, backend=None, supports=None) -> InfiniteModel:
    """``/root/reference/examples/quadrotor.jl:6-77`` — 9 states, 4 controls, T = 60,
    backward finite differences (InfiniteOpt default)."""
    n, p, T = 9, 4, 60.0
    im = InfiniteModel(backend)
    if supports is not None:   # explicit support window (shard.py)
        t = im.infinite_parameter("t", 0.0, T, supports=supports)
    else:
        t = im.infinite_parameter("t", 0.0, T, num_supports=num_supports)       # :19
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
    return im


def pandemic(num_supports: int = 100, num_scenarios: int = 4, backend=None, xi_supports=None) -> InfiniteModel:
    """``/root/reference/ESCAPE34/pandemic.jl:4-34`` — SIR optimal control.
    ξ supports are synthetic (equispaced in [0.1, 0.6]) instead of Julia-RNG
    ``Uniform(0.1, 0.6)`` draws (``:16``)."""
    gamma, beta, N = 0.303, 0.727, 1e5                                              # :8-10
    extra_ts = [0.001, 0.002, 0.004, 0.008, 0.02, 0.04, 0.08, 0.2, 0.4, 0.8]         # :11
    im = InfiniteModel(backend)
    t = im.infinite_parameter("t", 0.0, 200.0, num_supports=num_supports)            # :15
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
