#!/usr/bin/env python3
"""Is the placement mode of jac_coord! (the same kernel runs at 0.082-0.087 ms into some output buffers and at
0.094-0.097 ms into others) a property of THIS kernel's store pattern, or of the buffer?  For N output buffers of one
process: time jac_coord! into the buffer, a plain fill of the SAME bytes (torch's elementwise fill: one contiguous
stream, 16 bytes per lane) and hipMemsetAsync, each as the mean of `iters` back-to-back launches.  If the fill times
split the same way, no store pattern of ours can remove the mode — the buffer's physical pages decide for every writer.

  python tools/fill_vs_jac_probe.py [--supports 1000000] [--buffers 8] [--opt k=v ...]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel

ap = argparse.ArgumentParser()
ap.add_argument("--supports", type=int, default=1_000_000)
ap.add_argument("--buffers", type=int, default=8)
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--opt", action="append", default=[])
args = ap.parse_args()
opts = {k: int(v) for k, v in (kv.split("=") for kv in args.opt)}
core = transcribe.exa_core(workloads.quadrotor(args.supports))
gm = ExaModel(core, device=0, options=opts)
x = torch.tensor(gm.meta.x0 + 0.1 * np.random.default_rng(0).standard_normal(gm.meta.nvar), device="cuda")
y = torch.tensor(np.random.default_rng(1).standard_normal(gm.meta.ncon), device="cuda")
jbufs = [torch.empty(gm.meta.nnzj, dtype=torch.float64, device="cuda") for _ in range(args.buffers)]
hbufs = [torch.empty(gm.meta.nnzh, dtype=torch.float64, device="cuda") for _ in range(args.buffers)]


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters


rows = []
for rnd in range(2):
    for i, (jb, hb) in enumerate(zip(jbufs, hbufs)):
        r = dict(round=rnd, buffer=i, jac_addr=hex(jb.data_ptr()),
                 jac_ms=timed(lambda: gm.jac_coord(x, jb)), jac_fill_ms=timed(lambda: jb.fill_(1.0)), jac_memset_ms=timed(lambda: jb.zero_()),
                 hess_ms=timed(lambda: gm.hess_coord(x, y, hb)), hess_fill_ms=timed(lambda: hb.fill_(1.0)))
        rows.append(r)
        print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
j = np.array([r["jac_ms"] for r in rows]); f = np.array([r["jac_fill_ms"] for r in rows]); ms = np.array([r["jac_memset_ms"] for r in rows])
h = np.array([r["hess_ms"] for r in rows]); hf = np.array([r["hess_fill_ms"] for r in rows])
corr = lambda a, b: float(np.corrcoef(a, b)[0, 1]) if a.std() > 0 and b.std() > 0 else None
out = {"supports": args.supports, "options": opts, "bytes_jac": int(gm.meta.nnzj * 8), "bytes_hess": int(gm.meta.nnzh * 8), "rows": rows,
       "corr_jac_vs_fill": corr(j, f), "corr_jac_vs_memset": corr(j, ms), "corr_hess_vs_fill": corr(h, hf),
       "jac_ms_range": [float(j.min()), float(j.max())], "fill_ms_range": [float(f.min()), float(f.max())], "memset_ms_range": [float(ms.min()), float(ms.max())],
       "fill_GBps_range": [gm.meta.nnzj * 8 / f.max() / 1e6, gm.meta.nnzj * 8 / f.min() / 1e6]}
print(json.dumps(out))
