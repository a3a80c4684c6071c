"""Native vs foreign-producer encoding of one model on the GPU (profiles/r01_foreign_blob_perf.txt)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import numpy as np, torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel
from test_foreign_producer import foreign_blob
core = transcribe.exa_core(workloads.pandemic(4990, 100))
for tag, blob in (("native", core.to_blob()), ("foreign (flat iterators, lattice recovered by the library)", foreign_blob(core)[1])):
    gm = ExaModel.from_blob(blob, device=0)
    x = torch.tensor(np.abs(gm.meta.x0 + 0.1 * np.random.default_rng(0).standard_normal(gm.meta.nvar)) + 0.05, device="cuda")
    y = torch.tensor(np.random.default_rng(1).standard_normal(gm.meta.ncon), device="cuda")
    j = torch.empty(gm.meta.nnzj, dtype=torch.float64, device="cuda"); h = torch.empty(gm.meta.nnzh, dtype=torch.float64, device="cuda")
    ms_j, ms_h = gm.time_kernels(x, y, j, h, iters=100)
    print(f"pandemic 5000x100 {tag}: blob {len(blob) / 1e6:.0f} MB, jac {ms_j:.4f} ms, hess {ms_h:.4f} ms", flush=True)
    gm.close()
