#!/usr/bin/env python3
"""Does jac_coord!'s placement mode follow the START ADDRESS of the output buffer inside one allocation?
One arena of nnzj*8 bytes + slack; the window slides by `step` bytes; same kernel, same x, HIP-event timing (iem_time_kernels).
  python tools/slide_probe.py [supports] -> one line per offset"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel
import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
core = transcribe.exa_core(workloads.quadrotor(S))
gm = ExaModel(core, device=0, options={"autotune": 0})
x, y = bench.eval_point(gm.meta.nvar, gm.meta.ncon, gm.meta.x0, S, seed=0)
xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
hb = torch.empty(gm.meta.nnzh, dtype=torch.float64, device="cuda")
MiB = 1 << 20
for label, step, count in (("2MiB", 2 * MiB, 40), ("64KiB", 64 * 1024, 34), ("4KiB", 4096, 34), ("128B", 128, 34)):
    slack = step * count
    arena = torch.empty(gm.meta.nnzj + slack // 8 + 16, dtype=torch.float64, device="cuda")
    base = arena.data_ptr()
    row = []
    for k in range(count):
        off = k * step
        jb = arena[off // 8: off // 8 + gm.meta.nnzj]
        ms_j, _ = gm.time_kernels(xd, yd, jb, hb, iters=30)
        row.append(ms_j)
    print(f"step {label:6s} base 0x{base:x}: " + " ".join(f"{v:.4f}" for v in row), flush=True)
    del arena
    torch.cuda.empty_cache()
