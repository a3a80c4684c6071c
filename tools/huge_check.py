#!/usr/bin/env python3
"""Beyond 2^31 COO entries: quadrotor at 3.6e7 supports (nnzj = 2.23e9 > 2^31, 17.9 GB of Jacobian
values) — jac_coord!/hess_coord!/cons!/grad!/obj on the GPU against the FULL oracle (OpenMP on the
host cores).  Run once per round through gpurun; needs ~120 GB of host memory and ~60 GB of HBM."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel
from pyoracle import OracleModel

S = int(sys.argv[1]) if len(sys.argv) > 1 else 36_000_000
t0 = time.perf_counter()
core = transcribe.exa_core(workloads.quadrotor(S))
blob = core.to_blob()
t_build = time.perf_counter() - t0
gm = ExaModel(core, device=0, blob=blob, options={"autotune": 1})   # opt-in second code object: its store path beyond 2^31 too
print("model", gm.meta.nvar, gm.meta.ncon, gm.meta.nnzj, gm.meta.nnzh, "build %.1fs" % t_build, flush=True)
rng = np.random.default_rng(0)
x = gm.meta.x0 + 0.1 * rng.standard_normal(gm.meta.nvar)
x[7 * S:8 * S] = np.clip(x[7 * S:8 * S], -1.2, 1.2)
y = rng.standard_normal(gm.meta.ncon)
xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
res = {"supports": S, "nnzj": gm.meta.nnzj, "nnzh": gm.meta.nnzh, "nnzj_exceeds_int32": gm.meta.nnzj > 2**31 - 1}
om = OracleModel(blob)
om.set_threads(min(len(os.sched_getaffinity(0)), 16))


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


t0 = time.perf_counter(); jv = gm.jac_coord(xd); torch.cuda.synchronize(); res["jac_ms"] = (time.perf_counter() - t0) * 1e3
ref = om.jac_coord(x); jh = jv.cpu().numpy(); res["jac_rel_err"] = rel(jh, ref); del ref
# the store-batch tuner alternates the handle's two code objects over the first calls into a buffer: the SECOND call
# into the same buffer runs the large-batch object (positions beyond 2^31 through its store path too) — same bytes
jv.fill_(float("nan")); gm.jac_coord(xd, jv); torch.cuda.synchronize()
res["jac_second_code_object_identical"] = bool(np.array_equal(jv.cpu().numpy(), jh)); res["tuner_state_after_two_calls"] = gm.tuner_choice("jac", jv)
del jv, jh; torch.cuda.empty_cache()
print(res, flush=True)
t0 = time.perf_counter(); hv = gm.hess_coord(xd, yd, obj_weight=0.7); torch.cuda.synchronize(); res["hess_ms"] = (time.perf_counter() - t0) * 1e3
ref = om.hess_coord(x, y, 0.7); res["hess_rel_err"] = rel(hv.cpu().numpy(), ref); del hv, ref; torch.cuda.empty_cache()
res["cons_rel_err"] = rel(gm.cons(xd).cpu().numpy(), om.cons(x))
res["grad_rel_err"] = rel(gm.grad(xd).cpu().numpy(), om.grad(x))
vc = np.random.default_rng(3).standard_normal(gm.meta.ncon)
res["jtprod_rel_err"] = rel(gm.jtprod(xd, torch.tensor(vc, device="cuda")).cpu().numpy(), om.jtprod(x, vc))   # pulled stencil neighbours at this size
fo = om.obj(x); res["obj_rel_err"] = abs(gm.obj(xd) - fo) / max(1.0, abs(fo))
# structure: last entries (positions beyond 2^31) on the device vs the oracle
r, c = gm.jac_structure_device(1)
ro, co = om.jac_structure(1)
res["jac_structure_equal"] = bool(np.array_equal(r.cpu().numpy(), ro) and np.array_equal(c.cpu().numpy(), co))
print(json.dumps(res), flush=True)
