"""Worker of tests/test_gpu_comm.py: one process per rank, all ranks on cuda:0 (the one-GPU
rehearsal of the multi-GPU path).  Each rank owns only its slice of a DISTRIBUTED x: halo entries
arrive through iem_halo_exchange, the objective and the replicated gradient entries are summed by
iem_allreduce_obj_grad (mailboxes over HIP IPC, no torch.distributed on the data path — gloo only
moves the 128-byte handles and gathers results for checking)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

from infiniteexamodels.jl_amd import shard, transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel


def build_global(name, size):
    if name == "quadrotor":
        return transcribe.exa_core(workloads.quadrotor(size[0]))
    if name == "quadrotor_oc3":      # ESCAPE34 variant: OrthogonalCollocation(3), halo of TWO supports
        return transcribe.exa_core(workloads.quadrotor(size[0], collocation=3))
    if name == "farmer":
        return transcribe.exa_core(workloads.farmer(size[0]))
    if name == "opf":
        return transcribe.exa_core(workloads.opf(size[0]))
    return transcribe.exa_core(workloads.pandemic(size[0], size[1]))


def main():
    name, group = sys.argv[1], int(sys.argv[2])
    size = tuple(int(v) for v in sys.argv[3].split("x"))
    mode = sys.argv[4] if len(sys.argv) > 4 else "eager"
    use_graph = mode in ("graph", "async_graph")
    use_async = mode in ("async", "async_graph")     # iem_halo_exchange_async instead of the stream-ordered exchange
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    opts = {"split_small": 0}
    gcore = build_global(name, size)
    gblob = gcore.to_blob()
    gm = ExaModel.sharded(gblob, group, rank, world, device=0, options=opts)
    info = gm.shard_info()
    assert (info["rank"], info["world"]) == (rank, world)
    shard.connect_mailboxes(gm, dist)
    lay = shard.ShardLayout.of_model(gm)
    vm, halo, repl, owned = lay.var_map, lay.halo, lay.replicated, lay.owned
    row_map, jpos, hpos = lay.row_map, lay.jac_pos, lay.hess_pos
    if rank == 0:
        from pyoracle import OracleModel
        G = ExaModel(gcore, device=0, blob=gblob, options=opts)      # the unsharded model on the same GPU
        O = OracleModel(gblob)
    nvg, ncg = info["nvar_global"], info["ncon_global"]
    xd = torch.empty(gm.meta.nvar, dtype=torch.float64, device="cuda")
    yd = torch.empty(gm.meta.ncon, dtype=torch.float64, device="cuda")
    f = torch.zeros(1, dtype=torch.float64, device="cuda")
    g = torch.empty(gm.meta.nvar, dtype=torch.float64, device="cuda")
    c = torch.empty(gm.meta.ncon, dtype=torch.float64, device="cuda")
    jv = torch.empty(gm.meta.nnzj, dtype=torch.float64, device="cuda")
    hv = torch.empty(gm.meta.nnzh, dtype=torch.float64, device="cuda")
    fpre = torch.zeros(1, dtype=torch.float64, device="cuda")
    gpre = torch.empty(gm.meta.nvar, dtype=torch.float64, device="cuda")
    jt = torch.empty(gm.meta.nvar, dtype=torch.float64, device="cuda")
    vd = torch.empty(gm.meta.nvar, dtype=torch.float64, device="cuda")
    hp = torch.empty(gm.meta.nvar, dtype=torch.float64, device="cuda")
    jp = torch.empty(gm.meta.ncon, dtype=torch.float64, device="cuda")

    exchange = gm.halo_exchange_async if use_async else gm.halo_exchange
    reads = gm.halo_reads()
    if name == "quadrotor" and rank > 0:     # difference rows: cons! reads x_k[a_r - 1], their partials are item data
        assert reads["cons"][0] and not reads["jac"][0] and not reads["hess"][0] and not reads["obj"][0], reads
        assert reads["obj"][2] and reads["jac"][2] and reads["pair"][2] and not reads["cons"][2] and not reads["grad"][2], reads   # who can carry

    phase_it = [False]

    def loop():
        exchange(xd)
        if use_async and phase_it[0]:
            # the same evaluations through ONE LAUNCH PER SOLVER PHASE: the accepted-point launch (grad! + jac_coord! +
            # hess_coord!) carries the deferred exchange when it can (no halo entry touched, no shared-entry epilogue), else it
            # is flushed in front; the trial-point launch (obj + cons!) reads halo entries on ranks > 0
            gm.eval_accepted(xd, yd, g, jv, hv, obj_weight=0.7)
            fv, _ = gm.eval_trial(xd, c)
            f.fill_(fv)
        elif use_async:
            # solver order: obj carries the deferred exchange (one extra workgroup), grad! neither touches nor carries,
            # the fused pair and cons! find the halo entries in x
            gm.obj_device(xd, f); gm.grad(xd, g); gm.jac_hess_coord(xd, yd, jv, hv, obj_weight=0.7); gm.cons(xd, c)
        else:
            gm.cons(xd, c); gm.jac_coord(xd, jv); gm.hess_coord(xd, yd, hv, obj_weight=0.7)
            gm.obj_device(xd, f); gm.grad(xd, g)
        fpre.copy_(f); gpre.copy_(g)
        gm.allreduce_obj_grad(f, g)
        # J'v with v = y: each rank's rows, then the transposed halo exchange (what my first difference row owes to
        # x_k[a_r - 1] goes to the left neighbour) and the sum over the ranks of the replicated variables' entries
        gm.jtprod(xd, yd, jt)
        gm.halo_fold(jt)
        gm.allreduce_obj_grad(None, jt)
        # J v and H v with a DISTRIBUTED v: its halo copies arrive like x's; H v folds back like J'v
        exchange(vd)
        gm.jprod(xd, vd, jp)
        gm.hprod(xd, yd, vd, hp, obj_weight=0.7)
        gm.halo_fold(hp)
        gm.allreduce_obj_grad(None, hp)

    graph = None
    for it in range(5):
        rng = np.random.default_rng(100 + it)
        xg = np.concatenate([np.zeros(0), 0.3 + 0.1 * rng.standard_normal(nvg)])
        if name not in ("quadrotor", "quadrotor_oc3", "opf"):
            xg = np.abs(xg) + 0.05
        yg = np.random.default_rng(200 + it).standard_normal(ncg)
        xl = xg[vm].copy()
        xl[halo] = np.nan                                  # this rank does NOT hold its neighbour's values
        xd.copy_(torch.tensor(xl)); yd.copy_(torch.tensor(yg[row_map]))
        vg = np.random.default_rng(300 + it).standard_normal(nvg)
        vl = vg[vm].copy()
        vl[halo] = np.nan
        vd.copy_(torch.tensor(vl))
        for out in (c, jv, hv, g, jt, hp, jp):
            out.fill_(float("nan"))
        if use_async and not use_graph:
            # ORDERING RULE (include/iem.h): a collective never overtakes a deferred exchange.  grad! neither touches a halo
            # entry nor carries; iem_allreduce_obj_grad must flush the exchange FIRST on every rank — the halo entries are
            # in x after a plain device synchronisation (iem_synchronize / iem_comm_status, which flush too, are not called)
            xo = xd.clone()
            gm.halo_exchange_async(xo)
            if not gm.halo_reads()["grad"][0]:
                gm.grad(xo, g)
            gm.allreduce_obj_grad(f, g)
            torch.cuda.synchronize()
            assert np.array_equal(xo.cpu().numpy(), xg[vm]), "iem_allreduce_obj_grad overtook a deferred halo exchange"
            g.fill_(float("nan"))
        phase_it[0] = use_async and not use_graph and it % 2 == 1
        if use_graph and it >= 2:
            if graph is None:                              # iterations 0, 1 ran eagerly (warm-up); capture once, replay after
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    loop()
            graph.replay()
        else:
            loop()
        torch.cuda.synchronize()
        assert gm.comm_status() == 0, "a mailbox wait timed out"
        assert np.array_equal(xd.cpu().numpy(), xg[vm]), "halo entries did not arrive"
        res = dict(c=c.cpu().numpy(), j=jv.cpu().numpy(), h=hv.cpu().numpy(), fpre=fpre.item(), fpost=f.item(),
                   gpre=gpre.cpu().numpy()[repl], gpost=g.cpu().numpy()[repl], gown=g.cpu().numpy()[owned & ~repl],
                   jtown=jt.cpu().numpy()[owned & ~repl], jtrepl=jt.cpu().numpy()[repl], jthalo=jt.cpu().numpy()[halo],
                   hpown=hp.cpu().numpy()[owned & ~repl], hprepl=hp.cpu().numpy()[repl], jp=jp.cpu().numpy(),
                   own_idx=vm[owned & ~repl], repl_idx=vm[repl], row_map=row_map, jpos=jpos, hpos=hpos)
        allres = [None] * world if rank == 0 else None
        dist.gather_object(res, allres, dst=0)
        if rank == 0:
            xgd, ygd = torch.tensor(xg, device="cuda"), torch.tensor(yg, device="cuda")
            cg, jg, hg = (np.full(n, np.nan) for n in (G.meta.ncon, G.meta.nnzj, G.meta.nnzh))
            gg = np.full(nvg, np.nan)
            jtg = np.full(nvg, np.nan)
            hpg, jpg = np.full(nvg, np.nan), np.full(G.meta.ncon, np.nan)
            for r in allres:
                jtg[r["own_idx"]] = r["jtown"]; jtg[r["repl_idx"]] = r["jtrepl"]
                hpg[r["own_idx"]] = r["hpown"]; hpg[r["repl_idx"]] = r["hprepl"]; jpg[r["row_map"]] = r["jp"]
                assert not r["jthalo"].size or not np.any(r["jthalo"]), "halo copies are zeroed by the fold"
            for r in allres:
                cg[r["row_map"]] = r["c"]; jg[r["jpos"]] = r["j"]; hg[r["hpos"]] = r["h"]
                gg[r["own_idx"]] = r["gown"]
                gg[r["repl_idx"]] = r["gpost"]
            # the reassembled shard results ARE the one-GPU results, bit for bit
            assert np.array_equal(cg, G.cons(xgd).cpu().numpy()), "cons"
            assert np.array_equal(jg, G.jac_coord(xgd).cpu().numpy()), "jac"
            assert np.array_equal(hg, G.hess_coord(xgd, ygd, obj_weight=0.7).cpu().numpy()), "hess"
            # ... and agree with the oracle on the GLOBAL model
            for got, ref, what in ((cg, O.cons(xg), "cons"), (jg, O.jac_coord(xg), "jac"), (hg, O.hess_coord(xg, yg, 0.7), "hess"), (gg, O.grad(xg), "grad"), (jtg, O.jtprod(xg, yg), "jtprod"),
                                   (jpg, O.jprod(xg, vg), "jprod"), (hpg, O.hprod(xg, yg, vg, 0.7), "hprod")):
                scale = np.maximum(np.abs(ref), 1e-10 * max(1.0, np.abs(ref).max() if ref.size else 1.0))
                assert ref.size == 0 or (np.abs(got - ref) / scale).max() <= 1e-10, what
            # all-reduce: rank-order sums, identical bits on every rank
            fsum = allres[0]["fpre"]
            gsum = allres[0]["gpre"].copy()
            for r in allres[1:]:
                fsum = fsum + r["fpre"]
                gsum = gsum + r["gpre"]
            for r in allres:
                assert r["fpost"] == fsum and np.array_equal(r["gpost"], gsum), "all-reduce result differs from the rank-order sum"
            assert abs(fsum - O.obj(xg)) <= 1e-10 * max(1.0, abs(O.obj(xg)))
        dist.barrier()
    if rank == 0:
        print("OK", name, world, info["halo_doubles"], info["n_shared"], "graph" if graph is not None else "eager",
              "mailbox_kind", gm.shard_info()["mailbox_kind"])
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
