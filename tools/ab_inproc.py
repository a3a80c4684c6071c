#!/usr/bin/env python3
"""Box- and placement-internal A/B of generator knobs: every variant is built in THIS process and
timed into the SAME output buffers (jac_coord!'s time depends on the output allocation —
profiles/r01_output_placement_probe.txt), three rounds, alternating.

  python tools/ab_inproc.py "overlap=1" "overlap=0" "block=1024" ...      (knob=value[,knob=value])
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infiniteexamodels.jl_amd import lib as iemlib, transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel
import bench

S = int(os.environ.get("IEM_AB_SUPPORTS", 1_000_000))
WL = os.environ.get("IEM_AB_WORKLOAD", "quadrotor")
im = {"quadrotor": lambda: workloads.quadrotor(S), "quadrotor_oc3": lambda: workloads.quadrotor(S, collocation=3),
      "opf": lambda: workloads.opf(S), "farmer": lambda: workloads.farmer(S),
      "pandemic": lambda: workloads.pandemic(S // 100 - 10, 100)}[WL]()
core = transcribe.exa_core(im)
print("workload", WL, S, flush=True)
blob = core.to_blob()
variants = sys.argv[1:] or ["overlap=1", "overlap=0"]
models = []
for v in variants:
    kw = {k: int(x) for k, x in (kv.split("=") for kv in v.split(",") if kv)}
    with iemlib.options(**kw):
        models.append(ExaModel(core, device=0, blob=blob))
gm = models[0]
import numpy as np
x = gm.meta.x0 + 0.1 * np.random.default_rng(0).standard_normal(gm.meta.nvar)
if WL in ("pandemic", "farmer"):
    x = np.abs(x) + 0.05
y = np.random.default_rng(1).standard_normal(gm.meta.ncon)
xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
bufs = [(torch.empty(gm.meta.nnzj, dtype=torch.float64, device="cuda"), torch.empty(gm.meta.nnzh, dtype=torch.float64, device="cuda")) for _ in range(3)]
for rnd in range(3):
    for v, m in zip(variants, models):
        row = []
        for jb, hb in bufs:
            for _ in range(24):         # lets a handle's store-batch tuner decide for THIS buffer pair (it needs completed events)
                m.jac_coord(xd, jb); m.hess_coord(xd, yd, hb)
                torch.cuda.synchronize()
            ms_j, ms_h = m.time_kernels(xd, yd, jb, hb, iters=50)
            cell = f"{ms_j:.4f}/{ms_h:.4f}"
            if os.environ.get("IEM_AB_PAIRLOOP"):     # also the bench's pattern: jac, hess, jac, hess, ... with no synchronisation in between
                step = m.raw_pair(xd, yd, jb, hb, obj_weight=1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(10):
                    step()
                torch.cuda.synchronize(); e0.record()
                for _ in range(100):
                    step()
                e1.record(); torch.cuda.synchronize()
                cell += f" pair {e0.elapsed_time(e1) / 100:.4f}"
                if os.environ.get("IEM_AB_FUSED"):     # ... and the one-launch form (iem_jac_hess_coord)
                    fstep = m.raw_pair(xd, yd, jb, hb, obj_weight=1.0, fused=True)
                    for _ in range(10):
                        fstep()
                    torch.cuda.synchronize(); e0.record()
                    for _ in range(100):
                        fstep()
                    e1.record(); torch.cuda.synchronize()
                    cell += f" fused {e0.elapsed_time(e1) / 100:.4f}"
            row.append(cell)
        print(f"round {rnd} {v:28s} jac/hess ms into 3 buffer pairs: " + "  ".join(row), flush=True)
