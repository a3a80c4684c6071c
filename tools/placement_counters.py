#!/usr/bin/env python3
"""Output-placement probe WITH counters.  Six 2-MiB-aligned output windows inside
ONE allocation; jac_coord!/hess_coord! are launched `ITERS` times into each window in turn, the
average event time per window is printed as JSON lines.  Run it under `rocprofv3 --pmc ...`: the
counter CSV lists the dispatches in launch order, `tools/placement_summary.py` groups them by window and
puts each window's counters next to its time — same process, same physical pages.

  python3 tools/placement_counters.py [supports] [iters]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel
import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
NWIN = 6
gm = ExaModel(transcribe.exa_core(workloads.quadrotor(S)), device=0)
x, y = bench.eval_point(gm.meta.nvar, gm.meta.ncon, gm.meta.x0, S)
xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
stride_j = ((gm.meta.nnzj * 8 + (1 << 21) - 1) >> 21 << 21) // 8
stride_h = ((gm.meta.nnzh * 8 + (1 << 21) - 1) >> 21 << 21) // 8
arena_j = torch.empty(NWIN * stride_j, dtype=torch.float64, device="cuda")
arena_h = torch.empty(NWIN * stride_h, dtype=torch.float64, device="cuda")
for w in range(NWIN):
    j = arena_j[w * stride_j:w * stride_j + gm.meta.nnzj]
    h = arena_h[w * stride_h:w * stride_h + gm.meta.nnzh]
    ms_j, ms_h = gm.time_kernels(xd, yd, j, h, iters=ITERS)   # 3 warm-up pairs + ITERS timed pairs
    print(json.dumps({"window": w, "supports": S, "launches_per_window": ITERS + 3, "jac_ms": ms_j, "hess_ms": ms_h,
                      "jac_ptr": hex(j.data_ptr()), "hess_ptr": hex(h.data_ptr())}), flush=True)
