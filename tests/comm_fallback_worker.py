"""Worker of tests/test_shard.py::test_comm_fallback_gloo: one process per rank, gloo, NO GPU.  The handle is a stub built
from the device-free cut (lib.shard_blob) whose comm_export fails the way a rank without an IPC-exportable mailbox does
(or, in "forced" mode, is never asked): shard.ShardComm must fall back to torch.distributed on every rank, move the halo
doubles point to point and sum the objective + replicated gradient entries with ONE all-reduce.  Evaluation by the CPU
oracle — this tests the fallback logic, not the kernels."""
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

from infiniteexamodels.jl_amd import lib as iemlib, shard, transcribe, workloads
from pyoracle import OracleModel


class StubHandle:
    """What ShardComm needs of an ExaModel when the mailboxes are unavailable."""

    def __init__(self, cut, fail_export: bool):
        _, self._info, self._vm, self._vf, _ = cut
        self.meta = SimpleNamespace(nvar=int(self._info["nvar"]))
        self._fail = fail_export

    def comm_export(self):
        if self._fail:
            raise iemlib.IemError("libiem_hip error -2: could not allocate an IPC-exportable mailbox")
        return b"\0" * iemlib.COMM_HANDLE_BYTES

    def comm_connect(self, handles):
        raise iemlib.IemError("libiem_hip error -2: this runtime gave no fine-grained IPC memory for the mailboxes")

    def shard_var_map(self):
        return self._vm, self._vf

    def shard_info(self):
        return self._info


def main():
    name, size, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gcore = transcribe.exa_core({"quadrotor": workloads.quadrotor, "farmer": workloads.farmer, "opf": workloads.opf}[name](size))
    gblob = gcore.to_blob()
    cut = iemlib.shard_blob(gblob, 1, rank, world)
    # "export": only rank 1 cannot export — every rank must still end on the fallback; "connect": all export, mapping fails
    gm = StubHandle(cut, fail_export=(mode == "export" and rank == 1))
    comm = shard.ShardComm(gm, dist, force_fallback=(mode == "forced"))
    assert comm.kind == "rccl" and comm.why, (comm.kind, comm.why)
    G, L = OracleModel(gblob), OracleModel(cut[0])
    lay = shard.ShardLayout.of_cut(cut)
    xg = np.abs(G.x0 + 0.1 * np.random.default_rng(0).standard_normal(G.nvar)) + 0.05
    xl = xg[lay.var_map].copy()
    xl[lay.halo] = np.nan                       # this rank does not hold its neighbour's values
    x = torch.from_numpy(xl)
    comm.halo_exchange(x)
    assert np.array_equal(x.numpy(), xg[lay.var_map]), "halo entries did not arrive through the fallback"
    f = torch.tensor([L.obj(x.numpy())], dtype=torch.float64)
    g = torch.from_numpy(L.grad(x.numpy()))
    comm.allreduce_obj_grad(f, g)
    fref, gref = G.obj(xg), G.grad(xg)
    assert abs(f.item() - fref) <= 1e-12 * max(1.0, abs(fref)), (f.item(), fref)
    sel = (lay.owned | lay.replicated) & ~lay.halo
    np.testing.assert_allclose(g.numpy()[sel], gref[lay.var_map][sel], rtol=1e-13, atol=1e-13)
    # a second round (buffers are reused) and the cons! rows that need the halo
    np.testing.assert_allclose(L.cons(x.numpy()), G.cons(xg)[lay.row_map], rtol=1e-13, atol=1e-13)
    dist.barrier()
    if rank == 0:
        print("OK", comm.kind, comm.why[:60])
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
