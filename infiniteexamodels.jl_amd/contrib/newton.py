"""A device-resident Lagrange–Newton solver for the solver slot of ``ExaTranscriptionBackend``.

The reference hands its ``ExaModel`` to MadNLP (+ CUDSS) or Ipopt (``/root/reference/README.md:36-37``,
``ext/InfiniteExaModelsMadNLP.jl:41-66``); per iteration such a solver calls ``obj, grad!, cons!, jac_coord!, hess_coord!``
and factorises the augmented system once.  For EQUALITY-constrained models without variable bounds — the quadrotor and
hovercraft tracking problems of the reference's examples — that iteration is all there is to the method: Newton on the KKT
conditions ``∇f + J'y = 0, c(x) = c_E`` with the regularised system ``[H + δw I, J'; J, −δc I]``, the inertia correction
interior-point codes use (δw grows until the factorisation reports ``ncon`` negative pivots) and backtracking on the ℓ1
merit function (with that inertia the step is a descent direction for it), a second-order correction against the Maratos
effect.  Everything stays on the device: the five evaluation calls, the CSR assembly (``kkt.KKTSystem``), the chain KKT
factorisation and solve (``kkt_chain.ChainKKT``); the host sees a few scalars per iteration.

Models with bounds or inequality rows are REFUSED here (``ipm.InteriorPointSolver`` takes them); models
whose supports do not form a chain fall back to a dense factorisation when small, and are refused otherwise.

    backend = ExaTranscriptionBackend(LagrangeNewtonSolver(tol=1e-8), backend=MI355XBackend())
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Any, Dict, List

import numpy as np


@dataclass
class NewtonResult:
    solution: Any                 # x (device tensor)
    multipliers: Any              # y:  ∇f(x) + J(x)'y = 0
    objective: float
    iterations: int
    status: str                   # JSO names: "first_order" | "max_iter" | "small_step" | "max_time"
    kkt_residual: float
    elapsed_s: float
    history: List[Dict[str, Any]] = field(default_factory=list)


class _Dense:
    """models without a chain (finite-parameter problems): the assembled matrix as a dense one"""

    def __init__(self, kkt):
        self.kkt = kkt

    def load(self):
        import torch
        k = self.kkt
        rp, ci = k.rowptr.to(torch.int64), k.colind.to(torch.int64)
        rows = torch.repeat_interleave(torch.arange(k.n, device=k.vals.device), rp[1:] - rp[:-1])
        self.A = torch.zeros(k.n, k.n, dtype=torch.float64, device=k.vals.device).index_put_((rows, ci), k.vals, accumulate=True)
        return self

    def factor(self):
        return self

    def inertia(self):
        import torch
        ev = torch.linalg.eigvalsh(self.A)
        neg = int((ev < 0).sum().item())
        return self.kkt.n - neg, neg, int((ev.abs() <= 1e-14 * max(1.0, float(ev.abs().max().item()))).sum().item())

    def solve(self, rhs, refine=1, rtol=0.0):
        import torch
        x = torch.linalg.solve(self.A, rhs)
        for _ in range(1 if refine == "auto" else int(refine)):
            x = x + torch.linalg.solve(self.A, rhs - self.A @ x)
        return x


class LagrangeNewtonSolver:
    """``solver(model, x0, y0, **options) -> NewtonResult`` — the callable ``ExaTranscriptionBackend`` expects.  The
    backend's ``silent`` / time-limit settings reach it as ``print_level`` (0: quiet, 5: a line per iteration) and
    ``max_wall_time`` — Ipopt's names and defaults (``ext/InfiniteExaModelsIpopt.jl:5-7``)."""
    option_convention = "ipopt"

    def __init__(self, tol: float = 1e-8, max_iter: int = 50, delta_w: float = 1e-8, delta_c: float = 1e-10, refine="auto",
                 linear_rtol: float = 1e-7, dense_limit: int = 3000, log=None, print_level: int = 0, max_wall_time: float = 1.0e20):
        # (refine = "auto": a linear solve is refined only while its residual exceeds linear_rtol * |rhs| — an inexact Newton step)
        self.opt = dict(tol=tol, max_iter=max_iter, delta_w=delta_w, delta_c=delta_c, refine=refine, linear_rtol=linear_rtol,
                        dense_limit=dense_limit, log=log, print_level=print_level, max_wall_time=max_wall_time)

    def __call__(self, model, x0=None, y0=None, **options) -> NewtonResult:
        import torch
        from .. import lib as _lib
        from ..kkt import KKTSystem
        from ..kkt_chain import ChainKKT
        o = dict(self.opt); o.update(options)
        t_start = time.perf_counter()
        meta = model.meta
        n, m = meta.nvar, meta.ncon
        if np.isfinite(meta.lvar).any() or np.isfinite(meta.uvar).any() or not np.array_equal(meta.lcon, meta.ucon):
            raise _lib.IemError("LagrangeNewtonSolver: the model has variable bounds or inequality rows; this solver takes equality-constrained "
                                "models only (ipm.InteriorPointSolver takes bounds and inequality rows)")
        dev = model.device
        # The KKT structure of a model never changes (set_parameter! moves values only): the CSR plan, the chain layout and
        # their device buffers are kept ON the model between solves — a re-solve (src/infiniteopt_backend.jl:511-615) starts
        # iterating at once instead of repeating seconds of set-up.  model.close() releases them.
        cached = getattr(model, "_newton_linear", None)
        if cached is None:
            kkt = KKTSystem(model)
            try:
                lin = ChainKKT(kkt)
            except _lib.IemError:
                if kkt.n > int(o["dense_limit"]):
                    kkt.close()
                    raise
                lin = _Dense(kkt)
            model._newton_linear = (kkt, lin)
        else:
            kkt, lin = cached
        T = lambda a: a.to(dev, torch.float64).clone() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev).clone()
        x = T(meta.x0 if x0 is None else x0)
        y = torch.zeros(m, dtype=torch.float64, device=dev) if y0 is None else T(y0)
        ceq = T(meta.lcon)
        g, c, jtv = torch.empty_like(x), torch.empty(m, dtype=torch.float64, device=dev), torch.empty_like(x)
        jv = torch.empty(meta.nnzj, dtype=torch.float64, device=dev)
        hv = torch.empty(meta.nnzh, dtype=torch.float64, device=dev)

        # a maximisation problem (meta.minimize false: ExaCore(minimize = false), src/transform.jl:814-815) is the minimisation
        # of -f: the iteration runs on sgn * f — the inertia test asks for a minimiser of that — and the multipliers are
        # handed back for f itself (grad f + J'y = 0)
        sgn = 1.0 if bool(getattr(meta, "minimize", True)) else -1.0

        def residual(xx, yy):
            model.grad(xx, g); model.cons(xx, c); model.jtprod(xx, yy, jtv)
            return torch.cat([sgn * g + jtv, c - ceq])

        hist: List[Dict[str, Any]] = []
        status, nu = "max_iter", 1.0
        r = residual(x, y)                      # (leaves grad f(x) in g and c(x) in c)
        f = model.obj(x)
        rn = float(r.abs().max().item())
        it = 0
        for it in range(int(o["max_iter"]) + 1):
            rn = float(r.abs().max().item())
            hist.append(dict(iter=it, kkt_residual=rn, obj=f))
            if o["log"]:
                o["log"](hist[-1])
            elif int(o["print_level"]) >= 5:
                print(f"iter {it:3d}  objective {f: .8e}  kkt residual {rn:.3e}", flush=True)
            if rn <= o["tol"]:
                status = "first_order"
                break
            if it == int(o["max_iter"]):
                break
            if time.perf_counter() - t_start > float(o["max_wall_time"]):
                status = "max_time"
                break
            t0 = time.perf_counter()
            model.jac_hess_coord(x, y, jv, hv, obj_weight=sgn)
            gdx_base = (sgn * g).clone()        # grad of the minimised objective at x (g is overwritten by trial points)
            viol = float((c - ceq).abs().sum().item())
            # Inertia correction as in Ipopt / MadNLP: the factorisation reports the pivot signs; while they are not
            # (nvar, ncon, 0) the Hessian block is shifted by a growing delta_w and the system factorised again.  With that
            # inertia the step is a descent direction of the l1 merit function  phi = f + nu |c - c_E|_1  (nu above the new
            # multipliers), which the line search below backtracks on (Armijo); a full step is also taken when it reduces
            # the KKT residual (no Maratos stall next to the solution).  A search that fails down to 2^-20 does not end in a
            # zero step: the shift is raised tenfold — towards a gradient step — and the direction computed again.
            dw, tries, accepted, step, searches = float(o["delta_w"]), 0, False, 0.0, 0
            while not accepted and searches < 6:
                searches += 1
                while True:
                    kkt.assemble(hv, jv, None, dw, float(o["delta_c"]))
                    lin.load().factor()
                    pos, neg, doubtful = lin.inertia()
                    tries += 1
                    if (neg == m and doubtful == 0) or tries >= 16:
                        break
                    dw = max(1e-4, dw * 10.0)
                if not (neg == m and doubtful == 0):
                    break      # no usable factorisation within the permitted shifts: never a direction from stale factors (-> small_step)
                d = lin.solve(-r, refine=o["refine"], rtol=float(o["linear_rtol"]))
                dx, dy = d[:n], d[n:]
                # the weight follows the multipliers of THIS step — up at once, down by halves (the wild multipliers of the
                # first iterations would otherwise price every later step by its second-order constraint violation)
                target = 1.1 * float((y + dy).abs().max().item()) + 1e-8
                nu = target if target >= nu else max(target, 0.5 * nu)
                slope = float((gdx_base @ dx).item()) - nu * viol
                phi0 = sgn * f + nu * viol
                step = 1.0
                for k in range(21):
                    xt = x + step * dx
                    ft = model.obj(xt)
                    model.cons(xt, c)
                    phit = sgn * ft + nu * float((c - ceq).abs().sum().item())
                    if slope < 0.0 and phit <= phi0 + 1e-4 * step * slope:
                        accepted = True
                    elif k == 0 or slope >= 0.0:
                        # the full step refused by the merit function: the watchdog judges it by the KKT residual, then a
                        # second-order correction (the same factors, right-hand side (0; c(x + dx) - c_E)) is tried — the
                        # Maratos effect is exactly a good step whose constraint curvature the l1 term overprices
                        rt = residual(xt, y + step * dy)
                        accepted = float(rt.abs().max().item()) < (1.0 - 1e-4 * step) * rn
                        if not accepted and k == 0 and slope < 0.0:
                            model.cons(xt, c)
                            soc = lin.solve(-torch.cat([torch.zeros_like(x), c - ceq]), refine=o["refine"], rtol=float(o["linear_rtol"]))
                            xs = xt + soc[:n]
                            fs = model.obj(xs)
                            model.cons(xs, c)
                            if sgn * fs + nu * float((c - ceq).abs().sum().item()) <= phi0 + 1e-4 * slope:
                                dx, dy, accepted = dx + soc[:n], dy + soc[n:], True
                    if accepted:
                        break
                    step *= 0.5
                if not accepted:
                    dw = max(1e-4, dw * 10.0)
            if not accepted:
                status = "small_step"
                break
            x, y = x + step * dx, y + step * dy
            r = residual(x, y)
            f = model.obj(x)
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            hist[-1].update(step=step, iteration_ms=(time.perf_counter() - t0) * 1e3, inertia=(pos, neg, doubtful), delta_w=dw, factorisations=tries,
                            merit_weight=nu)
        return NewtonResult(solution=x, multipliers=sgn * y, objective=float(hist[-1]["obj"]), iterations=it, status=status, kkt_residual=rn,
                            elapsed_s=time.perf_counter() - t_start, history=hist)
