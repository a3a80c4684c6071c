"""Worker of tests/test_shard.py::test_obj_grad_allreduce_gloo_world2 (one process per rank,
gloo backend; evaluation by the CPU oracle — this is a test of the sharding + collective
logic, not of the kernels)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

from infiniteexamodels.jl_amd import shard, transcribe, workloads
from pyoracle import OracleModel


def main():
    name, size = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gdata = transcribe.ExaMappingData()
    if name == "farmer":
        gcore = transcribe.exa_core(workloads.farmer(size), gdata)
        core, _ = shard.farmer_shard(size, rank, world)
    else:
        gcore = transcribe.exa_core(workloads.quadrotor(size), gdata)
        core, _ = shard.quadrotor_shard(size, rank, world)
    G, L = OracleModel(gcore.to_blob()), OracleModel(core.to_blob())
    maps = shard.ShardMaps(core, gcore, core._shard_spec, core._shard_data, gdata)
    xg = np.abs(G.x0 + 0.1 * np.random.default_rng(0).standard_normal(G.nvar)) + 0.05
    x = xg[maps.var_map]
    g_local = torch.from_numpy(L.grad(x))
    shared = torch.from_numpy(np.nonzero(maps.replicated)[0])
    f, g_local = shard.allreduce_obj_grad(L.obj(x), g_local, shared, dist)
    fref, gref = G.obj(xg), G.grad(xg)
    assert abs(f - fref) <= 1e-12 * max(1.0, abs(fref)), (f, fref)
    # replicated entries now hold the global sum; owned entries are exact already
    sel = maps.var_owned | maps.replicated
    np.testing.assert_allclose(g_local.numpy()[sel], gref[maps.var_map][sel], rtol=1e-13, atol=1e-13)
    dist.barrier()
    if rank == 0:
        print("OK", f, int(shared.numel()))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
