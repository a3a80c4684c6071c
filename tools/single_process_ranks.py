#!/usr/bin/env python3
"""One process driving several shard handles (here all on cuda:0, one torch stream each): mailboxes of the
same process are connected by pointer (no IPC).  Does the one-shot all-reduce complete when the ranks'
kernels are launched one after the other from ONE host thread?  (It needs the streams to run concurrently.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel

W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = transcribe.exa_core(workloads.farmer(3000)).to_blob()
ms = [ExaModel.sharded(g, 1, r, W) for r in range(W)]
hs = b"".join(m.comm_export() for m in ms)
for m in ms:
    m.comm_connect(hs)
streams = [torch.cuda.Stream() for _ in range(W)]
xs, fs, gs = [], [], []
for r, m in enumerate(ms):
    x = torch.tensor(np.abs(m.meta.x0 + 0.1 * np.random.default_rng(r).standard_normal(m.meta.nvar)) + 0.05, device="cuda")
    xs.append(x); fs.append(torch.zeros(1, dtype=torch.float64, device="cuda")); gs.append(torch.empty(m.meta.nvar, dtype=torch.float64, device="cuda"))
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(20):
    for r, m in enumerate(ms):
        with torch.cuda.stream(streams[r]):
            m.halo_exchange(xs[r]); m.obj_device(xs[r], fs[r]); m.grad(xs[r], gs[r])
    pre = None
    for r, m in enumerate(ms):
        with torch.cuda.stream(streams[r]):
            m.allreduce_obj_grad(fs[r], gs[r])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("status", [m.comm_status() for m in ms], "f", [f.item() for f in fs], "wall per iteration ms", dt / 20 * 1e3, "mailbox kind", ms[0].shard_info()["mailbox_kind"])
assert all(m.comm_status() == 0 for m in ms) and len({f.item() for f in fs}) == 1
print("OK")
