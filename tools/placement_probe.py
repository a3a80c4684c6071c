#!/usr/bin/env python3
"""Does jac_coord!/hess_coord! time depend on WHICH allocation the output lands in?  Several output
buffers are alive at once (distinct physical backing); each is timed in turn, twice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel
import bench

S = 1_000_000
gm = ExaModel(transcribe.exa_core(workloads.quadrotor(S)), device=0)
x, y = bench.eval_point(gm.meta.nvar, gm.meta.ncon, gm.meta.x0, S)
xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
jb = [torch.empty(gm.meta.nnzj, dtype=torch.float64, device="cuda") for _ in range(6)]
hb = [torch.empty(gm.meta.nnzh, dtype=torch.float64, device="cuda") for _ in range(6)]
for rep in range(2):
    for i in range(6):
        ms_j, ms_h = gm.time_kernels(xd, yd, jb[i], hb[i], iters=50)
        print(f"rep {rep} buffer {i}: jac {ms_j:.4f} ms @ {hex(jb[i].data_ptr())}   hess {ms_h:.4f} ms @ {hex(hb[i].data_ptr())}", flush=True)

# second experiment: six output windows inside ONE allocation
del jb, hb
torch.cuda.empty_cache()
stride = ((gm.meta.nnzj * 8 + (1 << 21) - 1) >> 21 << 21) // 8
arena = torch.empty(6 * stride, dtype=torch.float64, device="cuda")
hbuf = torch.empty(gm.meta.nnzh, dtype=torch.float64, device="cuda")
for i in range(6):
    j = arena[i * stride:i * stride + gm.meta.nnzj]
    ms_j, ms_h = gm.time_kernels(xd, yd, j, hbuf, iters=50)
    print(f"one arena, window {i}: jac {ms_j:.4f} ms @ {hex(j.data_ptr())}", flush=True)
