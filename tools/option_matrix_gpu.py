#!/usr/bin/env python3
"""Every A/B knob of the scatter / product paths, on the GPU, against the oracle: the opt-out forms (atomics instead
of pulled neighbours / axis sums / plan-driven gather, loads at the head, shared entries by atomics) must stay correct
as long as they are documented as options.  Models: the small cases of the test suite (hiprtc compiles each variant)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import cases
from infiniteexamodels.jl_amd.model import ExaModel
from pyoracle import OracleModel

COMBOS = [{}, {"det_scatter": 2}, {"det_scatter": 0}, {"pull_scatter": 0, "det_scatter": 0}, {"det_axis": 0, "det_scatter": 0},
          {"det_shared": 0, "det_scatter": 0}, {"lazy_loads": 1, "lazy_min_loads": 1}, {"lazy_loads": 2, "lazy_min_loads": 1},
          {"pull_scatter": 0}, {"det_axis": 0}, {"split_small": 0, "det_scatter": 2}]
MODELS = ["quadrotor_100", "quadrotor_1000", "quadrotor_oc3_700", "pandemic_300x7", "farmer_1000", "opf_600", "hovercraft", "hovercraft_oc4",
          "kinetic_20", "test_problem_1_oc3", "irregular", "three_node_50", "pfun_full"]


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max())) if len(b) else 0.0


worst, n = 0.0, 0
for name in MODELS:
    core = cases.build_core(name)
    blob = core.to_blob()
    om = OracleModel(blob)
    x, y = cases.eval_point_for(name, om)
    rng = np.random.default_rng(5)
    v, vc = rng.standard_normal(om.nvar), rng.standard_normal(om.ncon)
    xd, yd, vd, vcd = (torch.tensor(a, device="cuda") for a in (x, y, v, vc))
    ref = dict(grad=om.grad(x), cons=om.cons(x), jac=om.jac_coord(x), hess=om.hess_coord(x, y, 0.7), jprod=om.jprod(x, v),
               jtprod=om.jtprod(x, vc), hprod=om.hprod(x, y, v, 0.7))
    for opts in COMBOS:
        gm = ExaModel(core, device=0, blob=blob, options=opts)
        nan = lambda k: torch.full((max(k, 1),), float("nan"), device="cuda", dtype=torch.float64)[:k]
        got = dict(grad=gm.grad(xd, nan(om.nvar)), cons=gm.cons(xd, nan(om.ncon)), jac=gm.jac_coord(xd), hess=gm.hess_coord(xd, yd, obj_weight=0.7),
                   jprod=gm.jprod(xd, vd, nan(om.ncon)), jtprod=gm.jtprod(xd, vcd, nan(om.nvar)), hprod=gm.hprod(xd, yd, vd, nan(om.nvar), obj_weight=0.7))
        for k, g in got.items():
            e = rel(g.cpu().numpy(), ref[k])
            worst = max(worst, e); n += 1
            assert e <= 1e-10, (name, opts, k, e)
        assert abs(gm.obj(xd) - om.obj(x)) <= 1e-10 * max(1.0, abs(om.obj(x))), (name, opts, "obj")
        gm.close()
    print(name, "ok", flush=True)
print(json.dumps({"models": len(MODELS), "option_sets": len(COMBOS), "comparisons": n, "worst_rel_err": worst}))
