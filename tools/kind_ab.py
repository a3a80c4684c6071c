#!/usr/bin/env python3
"""In-process A/B of generator knobs for ONE evaluation kind (obj, cons, grad, jac, hess): every variant built in this process,
timed in blocks of back-to-back launches between one event pair, alternating, three rounds.

  python tools/kind_ab.py cons "block=0" "block=256" "lds_slots=48" ...      (IEM_AB_WORKLOAD / IEM_AB_SUPPORTS as ab_inproc.py)
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from infiniteexamodels.jl_amd import lib as iemlib, transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel

kind = sys.argv[1]
variants = sys.argv[2:] or ["block=0"]
S = int(os.environ.get("IEM_AB_SUPPORTS", 1_000_000))
WL = os.environ.get("IEM_AB_WORKLOAD", "quadrotor")
im = {"quadrotor": lambda: workloads.quadrotor(S), "pandemic": lambda: workloads.pandemic(S // 100 - 10, 100)}[WL]()
core = transcribe.exa_core(im)
blob = core.to_blob()
models = []
for v in variants:
    kw = {k: int(x) for k, x in (kv.split("=") for kv in v.split(",") if kv)}
    with iemlib.options(**kw):
        models.append(ExaModel(core, device=0, blob=blob))
gm = models[0]
x = gm.meta.x0 + 0.1 * np.random.default_rng(0).standard_normal(gm.meta.nvar)
if WL == "pandemic":
    x = np.abs(x) + 0.05
y = np.random.default_rng(1).standard_normal(gm.meta.ncon)
xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
n_out = {"cons": gm.meta.ncon, "grad": gm.meta.nvar, "jac": gm.meta.nnzj, "hess": gm.meta.nnzh, "obj": 1}[kind]
bufs = [torch.empty(n_out, dtype=torch.float64, device="cuda") for _ in range(3)]
p = lambda a: C.c_void_p(a.data_ptr())


def call(m, out):
    L, h = m._L, m._h
    return {"obj": lambda: L.iem_obj_device(h, p(xd), p(out)), "cons": lambda: L.iem_cons(h, p(xd), p(out)), "grad": lambda: L.iem_grad(h, p(xd), p(out)), "jac": lambda: L.iem_jac_coord(h, p(xd), p(out)),
            "hess": lambda: L.iem_hess_coord(h, p(xd), p(yd), 1.0, p(out))}[kind]


print("workload", WL, S, "kind", kind, flush=True)
for rnd in range(3):
    for v, m in zip(variants, models):
        m._sync_stream()
        row = []
        for out in bufs:
            f = call(m, out)
            for _ in range(30):
                iemlib.check(f())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                f()
            e1.record(); torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / 200 * 1e3)
        print(f"round {rnd}  {v:40s} " + "  ".join(f"{u:7.2f}" for u in row) + " us", flush=True)
