"""Worker of tests/test_gpu_comm.py::test_a_skipped_exchange_surfaces_as_an_error: two ranks on cuda:0, rank 0
deliberately SKIPS one halo exchange.  Rank 1's exchange kernel must not hang (bounded wait, option comm_timeout_ms), must
not leave the halo entries stale (they become NaN), and the next host synchronisation point must return IEM_E_COMM once."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

from infiniteexamodels.jl_amd import lib as iemlib, shard, transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == 2
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    blob = transcribe.exa_core(workloads.quadrotor(3000)).to_blob()
    gm = ExaModel.sharded(blob, 1, rank, world, device=0, options={"split_small": 0, "comm_timeout_ms": 400})
    shard.connect_mailboxes(gm, dist)
    vm, vf = gm.shard_var_map()
    halo = (vf & 4) != 0
    x = torch.full((gm.meta.nvar,), 0.25, dtype=torch.float64, device="cuda")
    # round 1: a complete exchange — no error anywhere
    gm.halo_exchange(x)
    gm.synchronize()
    assert gm.comm_status() == 0
    dist.barrier()
    # round 2: rank 0 skips its call
    if rank == 1:
        x[torch.tensor(np.nonzero(halo)[0], device="cuda")] = 7.0      # stale values that must NOT survive
        t0 = time.perf_counter()
        gm.halo_exchange(x)
        c = gm.cons(x)                                                  # consumes the halo entries: must see the poison
        try:
            gm.synchronize()
            raise AssertionError("the time-out did not surface at the host synchronisation point")
        except iemlib.IemError as e:
            assert "error -6" in str(e) and "timed out" in str(e), str(e)
        dt = time.perf_counter() - t0
        assert 0.3 < dt < 3.0, f"bounded wait took {dt:.2f} s (comm_timeout_ms = 400)"
        xh = x.cpu().numpy()
        assert halo.any() and np.isnan(xh[halo]).all(), "halo entries must be poisoned, not stale"
        assert np.isfinite(xh[~halo]).all()
        assert torch.isnan(c).any(), "the rows that read the halo carry the poison"
        gm.synchronize()                                                # reported once, then cleared
        assert gm.comm_status() == 0
        # the objective's host sync point reports it too (a second time-out: rank 0 is still not taking part)
        gm.halo_exchange(x)
        try:
            gm.obj(x)
            raise AssertionError("iem_obj did not report the time-out")
        except iemlib.IemError as e:
            assert "error -6" in str(e)
    dist.barrier()
    if rank == 0:
        print("OK timeout surfaced")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
