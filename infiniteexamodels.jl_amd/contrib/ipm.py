"""A device-resident primal-dual interior-point solver for the solver slot of ``ExaTranscriptionBackend`` — models WITH
variable bounds and inequality rows (``newton.LagrangeNewtonSolver`` takes the equality-constrained ones).

The reference hands its ``ExaModel`` to MadNLP / Ipopt (``/root/reference/README.md:36-37``, ``ext/*.jl``); this is the same
class of method, restated compactly from the published algorithm of Ipopt (Wächter & Biegler 2006: monotone barrier update,
fraction-to-the-boundary rule, inertia-corrected reduced KKT systems, the filter line search — and, there being no
restoration phase, the ℓ1 merit function of ``newton.py`` on the barrier problem whenever the filter finds no step;
``line_search = "merit"`` uses that one alone).  Per iteration: the five evaluation calls of the hot path, one assembly of

        [ H + Σx + δw I     J'   ] [dx]     [ ∇f + J'y − μ/(x − l) + μ/(u − x)        ]
        [       J          −D    ] [dy] = − [ c − c_E   |   c − s + Σs⁻¹ r_s           ]

(``D = δc`` on equality rows, ``Σs⁻¹`` on inequality rows whose slacks ``s`` are eliminated) and its factorisation by the
chain KKT solver (``kkt_chain.ChainKKT``: pivot signs → inertia) or, for small models without a chain, a dense one.  Everything
stays on the device; the host sees a handful of scalars per iteration.

    backend = ExaTranscriptionBackend(InteriorPointSolver(tol=1e-8), backend=MI355XBackend())
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional

import numpy as np


@dataclass
class IPMResult:
    solution: Any                 # x
    multipliers: Any              # y:  ∇f + J'y − zL + zU = 0   (NLPModels sign)
    multipliers_L: Any            # zL >= 0 (lower bounds of x)
    multipliers_U: Any            # zU >= 0
    objective: float
    iterations: int
    status: str                   # JSO names: "first_order" | "acceptable" | "max_iter" | "small_step" | "max_time" | "infeasible"
    kkt_residual: float
    elapsed_s: float
    history: List[Dict[str, Any]] = field(default_factory=list)


class InteriorPointSolver:
    """``solver(model, x0, y0, **options) -> IPMResult``.  Options under Ipopt's names where they exist (``tol``,
    ``max_iter``, ``mu_init``, ``print_level``, ``max_wall_time``, ``bound_push``, ``bound_relax_factor``)."""
    option_convention = "ipopt"

    def __init__(self, linear: Optional[Callable] = None, **options):
        self.linear_factory = linear          # model -> linear-system object (tests inject a host one); None: the device's
        self.opt = dict(tol=1e-8, acceptable_tol=1e-6, acceptable_iter=15, max_iter=300, mu_init=0.1, mu_min=1e-11, kappa_eps=10.0, kappa_mu=0.2, theta_mu=1.5,
                        tau_min=0.99, bound_push=1e-2, bound_frac=1e-2, bound_relax_factor=1e-8, delta_w=0.0, delta_c=1e-10,
                        refine="auto", linear_rtol=1e-8, dense_limit=3000, print_level=0, max_wall_time=1.0e20, log=None, mu_from_start=False,
                        line_search="filter", nlp_scaling_max_gradient=100.0)
        self.opt.update(options)

    # ------------------------------------------------------------------------------------------------------------
    def __call__(self, model, x0=None, y0=None, **options) -> IPMResult:
        import torch
        from .. import lib as _lib
        o = dict(self.opt); o.update(options)
        t_start = time.perf_counter()
        meta = model.meta
        n, m = int(meta.nvar), int(meta.ncon)
        dev = model.device
        f64 = torch.float64
        T = lambda a: a.to(dev, f64).clone() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev).clone()
        sgn = 1.0 if bool(getattr(meta, "minimize", True)) else -1.0
        lin = self.linear_factory(model) if self.linear_factory is not None else _device_linear(model, int(o["dense_limit"]))
        scaled = None
        if float(o["nlp_scaling_max_gradient"]) > 0.0 and n:
            scaled = model = _Scaled(model, T(meta.x0 if x0 is None else x0), float(o["nlp_scaling_max_gradient"]))
            meta = model.meta
        lvar, uvar, lcon, ucon = (np.asarray(a, dtype=np.float64) for a in (meta.lvar, meta.uvar, meta.lcon, meta.ucon))
        relax = float(o["bound_relax_factor"])      # (a fixed variable, lvar == uvar, becomes an interval of twice this width)
        rl = lambda b: b - relax * np.maximum(1.0, np.abs(b))      # Ipopt relaxes every finite bound a little (bound_relax_factor)
        ru = lambda b: b + relax * np.maximum(1.0, np.abs(b))
        eq = lcon == ucon
        ine = ~eq
        mI = int(ine.sum())
        xl, xu = T(np.where(np.isfinite(lvar), rl(lvar), -np.inf)), T(np.where(np.isfinite(uvar), ru(uvar), np.inf))
        sl, su = T(np.where(np.isfinite(lcon[ine]), rl(lcon[ine]), -np.inf)), T(np.where(np.isfinite(ucon[ine]), ru(ucon[ine]), np.inf))
        Lx, Ux, Ls, Us = torch.isfinite(xl), torch.isfinite(xu), torch.isfinite(sl), torch.isfinite(su)
        ceq = T(np.where(eq, lcon, 0.0))
        ine_idx = torch.as_tensor(np.nonzero(ine)[0], device=dev, dtype=torch.int64)
        eq_mask = torch.as_tensor(eq, device=dev)

        def push(v, lo, hi, has_lo, has_hi):      # the start point strictly inside (Ipopt: bound_push / bound_frac)
            k1, k2 = float(o["bound_push"]), float(o["bound_frac"])
            both = has_lo & has_hi
            span = torch.where(both, hi - lo, torch.full_like(v, float("inf")))
            pl = torch.minimum(k1 * torch.clamp(lo.abs(), min=1.0), k2 * span)
            pu = torch.minimum(k1 * torch.clamp(hi.abs(), min=1.0), k2 * span)
            v = torch.where(has_lo, torch.maximum(v, lo + pl), v)
            return torch.where(has_hi, torch.minimum(v, hi - pu), v)

        x = push(T(meta.x0 if x0 is None else x0), xl, xu, Lx, Ux)
        g, c, jtv = torch.empty(n, dtype=f64, device=dev), torch.empty(m, dtype=f64, device=dev), torch.empty(n, dtype=f64, device=dev)
        jv, hv = torch.empty(int(meta.nnzj), dtype=f64, device=dev), torch.empty(int(meta.nnzh), dtype=f64, device=dev)
        model.cons(x, c)
        s = push(c[ine_idx].clone(), sl, su, Ls, Us)
        y = torch.zeros(m, dtype=f64, device=dev) if y0 is None else sgn * T(y0)
        if scaled is not None and y0 is not None:
            y = y * scaled.df / scaled.dc
        zL, zU = torch.where(Lx, torch.ones_like(x), torch.zeros_like(x)), torch.where(Ux, torch.ones_like(x), torch.zeros_like(x))
        vL, vU = torch.where(Ls, torch.ones_like(s), torch.zeros_like(s)), torch.where(Us, torch.ones_like(s), torch.zeros_like(s))
        mu = float(o["mu_init"])
        big = torch.full_like(x, float("inf"))
        bigs = torch.full_like(s, float("inf"))
        if o.get("mu_from_start", False):
            # a start point far inside its bounds (gaps of hundreds against multipliers of one) is far off the central path of
            # mu_init: steps then die on the fraction-to-the-boundary rule (the two-stage LP of examples/2stage_example.jl
            # crawled at steps of 1e-3).  Start the barrier parameter at the point's own average complementarity instead.
            prods = torch.cat([(x - xl)[Lx], (xu - x)[Ux], (s - sl)[Ls], (su - s)[Us]])
            if prods.numel():
                mu = min(max(mu, float(prods.mean().item())), 1e4)

        def gaps(xx, ss):
            return (torch.where(Lx, xx - xl, big), torch.where(Ux, xu - xx, big), torch.where(Ls, ss - sl, bigs), torch.where(Us, su - ss, bigs))

        def barrier(xx, ss):
            dL, dU, eL, eU = gaps(xx, ss)
            t = torch.log(dL[Lx]).sum() + torch.log(dU[Ux]).sum() + torch.log(eL[Ls]).sum() + torch.log(eU[Us]).sum()
            return float(t.item())

        def infeasibility(cc, ss):
            r = cc - ceq
            r[ine_idx] = cc[ine_idx] - ss
            return r

        def evaluate(xx, yy):
            """f, grad (in g), cons (in c), J'y (in jtv) at xx"""
            model.grad(xx, g); model.cons(xx, c); model.jtprod(xx, yy, jtv)
            return model.obj(xx)

        def errors(mu_):
            """Ipopt's optimality error E_mu (scaled): dual infeasibility, primal infeasibility, complementarity"""
            dL, dU, eL, eU = gaps(x, s)
            rd = sgn * g + jtv - zL + zU
            rs = -y[ine_idx] - vL + vU
            rp = infeasibility(c, s)
            comp = torch.cat([(dL * zL - mu_)[Lx], (dU * zU - mu_)[Ux], (eL * vL - mu_)[Ls], (eU * vU - mu_)[Us]])
            nz = int(Lx.sum() + Ux.sum() + Ls.sum() + Us.sum())
            zs = float((zL.sum() + zU.sum() + vL.sum() + vU.sum()).item())
            sd = max(100.0, (float(y.abs().sum().item()) + zs) / max(1, m + nz)) / 100.0
            sc = max(100.0, zs / max(1, nz)) / 100.0
            ed = max(float(rd.abs().max().item()), float(rs.abs().max().item()) if mI else 0.0) / sd
            ep = float(rp.abs().max().item()) if m else 0.0
            ec = float(comp.abs().max().item()) / sc if nz else 0.0
            return max(ed, ep, ec), ed, ep, ec

        hist: List[Dict[str, Any]] = []
        status, nu, it, acceptable_run = "max_iter", 1.0, 0, 0
        filter_mode = str(o["line_search"]) == "filter"
        filt: List[Any] = []          # (theta, phi) pairs no later iterate of this barrier problem may be dominated by
        theta_ref = None
        f = evaluate(x, y)
        e0 = errors(0.0)[0]
        for it in range(int(o["max_iter"]) + 1):
            e0, ed, ep, ec = errors(0.0)
            hist.append(dict(iter=it, obj=f / (scaled.df if scaled is not None else 1.0), kkt_residual=e0, dual_inf=ed, primal_inf=ep, compl=ec, mu=mu))
            if o["log"]:
                o["log"](hist[-1])
            elif int(o["print_level"]) >= 5:
                print(f"iter {it:3d}  objective {hist[-1]['obj']: .8e}  inf_pr {ep:.2e}  inf_du {ed:.2e}  compl {ec:.2e}  mu {mu:.1e}", flush=True)
            if e0 <= float(o["tol"]):
                status = "first_order"
                break
            acceptable_run = acceptable_run + 1 if e0 <= float(o["acceptable_tol"]) else 0
            if acceptable_run >= int(o["acceptable_iter"]):      # Ipopt's acceptable termination: 15 iterations in a row below acceptable_tol
                status = "acceptable"
                break
            if it == int(o["max_iter"]):
                status = "acceptable" if e0 <= float(o["acceptable_tol"]) else "max_iter"
                break
            if time.perf_counter() - t_start > float(o["max_wall_time"]):
                status = "max_time"
                break
            # barrier update (Fiacco-McCormick, monotone): as long as the barrier problem is solved to kappa_eps * mu
            mu_floor = max(float(o["mu_min"]), float(o["tol"]) / 10.0)
            while mu > mu_floor and errors(mu)[0] <= float(o["kappa_eps"]) * mu:
                mu = max(mu_floor, min(float(o["kappa_mu"]) * mu, mu ** float(o["theta_mu"])))
                filt = []              # a new barrier problem: the filter starts again
            tau = max(float(o["tau_min"]), 1.0 - mu)
            t0 = time.perf_counter()
            model.jac_hess_coord(x, y, jv, hv, obj_weight=sgn)
            dL, dU, eL, eU = gaps(x, s)
            sig_x = zL / dL + zU / dU                                    # (inf gaps give 0)
            sig_s = vL / eL + vU / eU
            gx = sgn * g - torch.where(Lx, mu / dL, torch.zeros_like(x)) + torch.where(Ux, mu / dU, torch.zeros_like(x))     # barrier gradient in x
            gs = -torch.where(Ls, mu / eL, torch.zeros_like(s)) + torch.where(Us, mu / eU, torch.zeros_like(s))              # ... in s
            r_s = gs - y[ine_idx]
            rp = infeasibility(c, s)
            viol = float(rp.abs().sum().item())
            dcv = torch.full((m,), float(o["delta_c"]), dtype=f64, device=dev)
            if mI:
                dcv[ine_idx] = 1.0 / torch.clamp(sig_s, min=1e-300)
            rc = rp.clone()
            if mI:
                rc[ine_idx] = rp[ine_idx] + r_s / torch.clamp(sig_s, min=1e-300)
            rhs = -torch.cat([gx + jtv, rc])
            dw, tries, accepted, step, searches = float(o["delta_w"]), 0, False, 0.0, 0
            phi_f = sgn * f - mu * barrier(x, s)
            while not accepted and searches < 8:
                searches += 1
                while True:
                    lin.assemble(hv, jv, sig_x, dw, dcv)
                    lin.factor()
                    pos, neg, doubtful = lin.inertia()
                    tries += 1
                    if (neg == m and doubtful == 0) or tries >= 24:
                        break
                    dw = 1e-4 if dw == 0.0 else dw * (100.0 if tries <= 2 else 8.0)
                if not (neg == m and doubtful == 0):
                    break      # no usable factorisation within the permitted shifts: never a direction from stale factors (-> small_step)
                d = lin.solve(rhs, refine=o["refine"], rtol=float(o["linear_rtol"]))
                dx, dy = d[:n], d[n:]
                ds = (dy[ine_idx] - r_s) / torch.clamp(sig_s, min=1e-300) if mI else s.clone()
                # fraction to the boundary: primal step for (x, s), dual step for the bound multipliers
                def ftb(v, dv, gap_lo, gap_hi):
                    a = 1.0
                    neg_ = dv < 0
                    if bool(neg_.any()):
                        a = min(a, float((-tau * gap_lo[neg_] / dv[neg_]).min().item()))
                    pos_ = dv > 0
                    if bool(pos_.any()):
                        a = min(a, float((tau * gap_hi[pos_] / dv[pos_]).min().item()))
                    return a
                a_max = min(ftb(x, dx, dL, dU), ftb(s, ds, eL, eU) if mI else 1.0)
                dzL = torch.where(Lx, mu / dL - zL - zL / dL * dx, torch.zeros_like(x))
                dzU = torch.where(Ux, mu / dU - zU + zU / dU * dx, torch.zeros_like(x))
                dvL = torch.where(Ls, mu / eL - vL - vL / eL * ds, torch.zeros_like(s))
                dvU = torch.where(Us, mu / eU - vU + vU / eU * ds, torch.zeros_like(s))
                a_z = 1.0
                for z_, dz_ in ((zL, dzL), (zU, dzU), (vL, dvL), (vU, dvU)):
                    neg_ = dz_ < 0
                    if bool(neg_.any()):
                        a_z = min(a_z, float((-tau * z_[neg_] / dz_[neg_]).min().item()))
                target = 1.1 * float((y + dy).abs().max().item()) + 1e-8 if m else 0.0
                nu = target if target >= nu else max(target, 0.5 * nu)
                slope = float((gx @ dx).item()) + (float((gs @ ds).item()) if mI else 0.0) - nu * viol
                phi0 = phi_f + nu * viol
                # (a filter search that finds no step falls back to the merit function for this iteration — there is no
                # restoration phase to send it to)
                for use_filter in ((True, False) if filter_mode else (False,)):
                    if accepted:
                        break
                    step = a_max
                    for k in range(30):
                        xt, st = x + step * dx, s + step * ds
                        ft = model.obj(xt)
                        model.cons(xt, c)
                        vt = float(infeasibility(c, st).abs().sum().item())
                        phibt = sgn * ft - mu * barrier(xt, st)
                        phit = phibt + nu * vt
                        if use_filter:
                            # Ipopt's filter (Waechter & Biegler 2006, section 2.3) on (theta = |infeasibility|_1, phi = barrier objective):
                            # an f-type step (switching condition) has to pass Armijo on phi, any other one has to improve theta or
                            # phi over the current point; nothing dominated by a filter entry is taken
                            if theta_ref is None:
                                theta_ref = max(1.0, viol)
                            slope_b = slope + nu * viol
                            ok = np.isfinite(phibt) and vt <= 1e4 * theta_ref and not any(vt >= th and phibt >= ph for th, ph in filt)
                            ftype = slope_b < 0.0 and viol <= 1e-4 * theta_ref and step * (-slope_b) ** 2.3 > viol ** 1.1
                            if ok and ftype:
                                accepted = phibt <= phi_f + 1e-8 * step * slope_b + 10.0 * np.finfo(float).eps * abs(phi_f)
                            elif ok:
                                accepted = vt <= (1.0 - 1e-5) * viol or phibt <= phi_f - 1e-5 * viol
                                if accepted:
                                    filt.append(((1.0 - 1e-5) * viol, phi_f - 1e-5 * viol))
                        elif np.isfinite(phit) and phit <= phi0 + 1e-8 * step * slope + 10.0 * np.finfo(float).eps * abs(phi0):
                            accepted = True
                        if not accepted and not use_filter and k == 0 and slope < 0.0 and np.isfinite(phit):
                            # second-order correction (the factors of this iteration, right-hand side (0; infeasibility at the trial point))
                            rpt = infeasibility(c, st)
                            soc = lin.solve(-torch.cat([torch.zeros_like(x), rpt]), refine=o["refine"], rtol=float(o["linear_rtol"]))
                            dx2 = dx * step + soc[:n]
                            ds2 = ds * step + (soc[n:][ine_idx] / torch.clamp(sig_s, min=1e-300) if mI else ds * 0)
                            a2 = min(ftb(x, dx2, dL, dU), ftb(s, ds2, eL, eU) if mI else 1.0)
                            xs_, ss_ = x + a2 * dx2, s + a2 * ds2
                            fs = model.obj(xs_)
                            model.cons(xs_, c)
                            phis = sgn * fs - mu * barrier(xs_, ss_) + nu * float(infeasibility(c, ss_).abs().sum().item())
                            if np.isfinite(phis) and phis <= phi0 + 1e-8 * a2 * slope:
                                dx, ds, dy, step, accepted = dx2, ds2, dy * step + soc[n:], a2, True
                        if accepted:
                            break
                        step *= 0.5
                if not accepted:
                    dw = 1e-4 if dw == 0.0 else dw * 10.0
            if not accepted:
                status = "small_step"
                break
            x, s, y = x + step * dx, s + step * ds, y + step * dy
            zL, zU, vL, vU = zL + a_z * dzL, zU + a_z * dzU, vL + a_z * dvL, vU + a_z * dvU
            # keep the bound multipliers within kappa_sigma of mu / gap (Ipopt eq. 16)
            dL, dU, eL, eU = gaps(x, s)
            ks = 1e10
            zL = torch.where(Lx, torch.minimum(torch.maximum(zL, mu / (ks * dL)), ks * mu / dL), zL)
            zU = torch.where(Ux, torch.minimum(torch.maximum(zU, mu / (ks * dU)), ks * mu / dU), zU)
            vL = torch.where(Ls, torch.minimum(torch.maximum(vL, mu / (ks * eL)), ks * mu / eL), vL)
            vU = torch.where(Us, torch.minimum(torch.maximum(vU, mu / (ks * eU)), ks * mu / eU), vU)
            f = evaluate(x, y)
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            hist[-1].update(step=step, step_dual=a_z, iteration_ms=(time.perf_counter() - t0) * 1e3, inertia=(pos, neg, doubtful), delta_w=dw,
                            factorisations=tries, merit_weight=nu)
        if scaled is not None:
            y, zL, zU = scaled.unscale(y, zL, zU)
            f = f / scaled.df
        return IPMResult(solution=x, multipliers=sgn * y, multipliers_L=sgn * zL, multipliers_U=sgn * zU,
                         objective=float(f), iterations=it, status=status, kkt_residual=float(e0), elapsed_s=time.perf_counter() - t_start, history=hist)


class _Scaled:
    """Gradient-based scaling (Ipopt's ``nlp_scaling_method = gradient-based``): the objective and every constraint row are
    multiplied by ``min(1, max_gradient / largest entry of its gradient at the start point)`` — a population of 1e5 people or a
    rate constant of 1e12 otherwise decide the step lengths.  The solver iterates on the scaled functions; ``unscale`` turns
    its multipliers into those of the model as stated."""

    def __init__(self, model, x0, max_gradient: float):
        import torch
        self.inner, self.device = model, model.device
        meta = model.meta
        g = model.grad(x0)
        gmax = float(g.abs().max().item()) if g.numel() else 0.0
        self.df = min(1.0, max_gradient / gmax) if gmax > 0.0 else 1.0
        m = int(meta.ncon)
        self.dc = torch.ones(m, dtype=torch.float64, device=self.device)
        if m and int(meta.nnzj):
            rows = model.jac_structure_device(0)[0]
            jv = torch.empty(int(meta.nnzj), dtype=torch.float64, device=self.device)
            hv = torch.empty(int(meta.nnzh), dtype=torch.float64, device=self.device)
            model.jac_hess_coord(x0, torch.zeros(m, dtype=torch.float64, device=self.device), jv, hv, obj_weight=0.0)
            rowmax = torch.zeros(m, dtype=torch.float64, device=self.device).scatter_reduce(0, rows, jv.abs(), reduce="amax")
            self.dc = torch.where(rowmax > max_gradient, max_gradient / rowmax, torch.ones_like(rowmax))
            self.rows = rows
        dcn = self.dc.cpu().numpy()
        self.meta = type("ScaledMeta", (), dict(nvar=meta.nvar, ncon=meta.ncon, nnzj=meta.nnzj, nnzh=meta.nnzh, x0=meta.x0, lvar=meta.lvar,
                                                uvar=meta.uvar, lcon=np.asarray(meta.lcon) * dcn, ucon=np.asarray(meta.ucon) * dcn,
                                                minimize=getattr(meta, "minimize", True)))()

    def obj(self, x):
        return self.df * self.inner.obj(x)

    def grad(self, x, g=None):
        g = self.inner.grad(x, g)
        return g.mul_(self.df)

    def cons(self, x, c=None):
        c = self.inner.cons(x, c)
        return c.mul_(self.dc)

    def jtprod(self, x, v, out=None):
        return self.inner.jtprod(x, v * self.dc, out)

    def jac_hess_coord(self, x, y, jac=None, hess=None, obj_weight: float = 1.0):
        self.inner.jac_hess_coord(x, y * self.dc, jac, hess, obj_weight=obj_weight * self.df)
        jac.mul_(self.dc[self.rows])
        return jac, hess

    def unscale(self, y, zL, zU):
        return y * self.dc / self.df, zL / self.df, zU / self.df


class _DeviceLinear:
    """``kkt.KKTSystem`` (CSR assembly on the device) + ``kkt_chain.ChainKKT`` or the dense fallback, kept on the model
    between solves (shared with ``newton.LagrangeNewtonSolver``)."""

    def __init__(self, model, dense_limit: int):
        from .. import lib as _lib
        from ..kkt import KKTSystem
        from ..kkt_chain import ChainKKT
        from .newton import _Dense
        cached = getattr(model, "_newton_linear", None)
        if cached is None:
            kkt = KKTSystem(model)
            try:
                lin = ChainKKT(kkt)
            except _lib.IemError:
                if kkt.n > dense_limit:
                    kkt.close()
                    raise
                lin = _Dense(kkt)
            model._newton_linear = cached = (kkt, lin)
        self.kkt, self.lin = cached

    def assemble(self, hv, jv, sigma, dw, dc):
        self.kkt.assemble(hv, jv, sigma, dw, dc)
        self.lin.load()

    def factor(self):
        self._singular = False
        try:
            self.lin.factor()
        except RuntimeError:      # the border's Schur complement did not factorise (an exactly singular system): report it as
            self._singular = True   # doubtful pivots, the caller shifts and factorises again

    def inertia(self):
        return (0, 0, 1) if self._singular else self.lin.inertia()

    def solve(self, rhs, refine="auto", rtol=1e-8):
        return self.lin.solve(rhs, refine=refine, rtol=rtol)


def _device_linear(model, dense_limit: int) -> _DeviceLinear:
    return _DeviceLinear(model, dense_limit)
