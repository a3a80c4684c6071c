#!/usr/bin/env python3
"""Multi-GPU full evaluation loop (BASELINE.json configs 3-5: pandemic, 8-GPU OPF, 8-GPU 2-stage):
obj, grad!, cons!, jac_coord!, hess_coord! on a support/scenario-sharded model, one process per
GPU, with the ONE data-path collective of the design — the all-reduce of the objective and of the
gradient entries of replicated (first-stage / non-sharded) variables.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      tools/eval_loop_dist.py --workload farmer|opf|pandemic|quadrotor --supports S_PER_RANK

Weak scaling: every rank owns S_PER_RANK scenarios (supports) of an N*S_PER_RANK problem.  Every
rank builds the GLOBAL blob and iem_create_sharded cuts its window (the C-ABI path a Julia host
would take).  The collective is the library's own one-shot mailbox all-reduce by default
(`--allreduce own`: iem_allreduce_obj_grad, preceded by iem_halo_exchange where a stencil crosses
the shard boundary); `--allreduce rccl` runs torch.distributed's instead, for comparison.
Rehearsal on ONE GPU: `--gpus N --same-device` spawns N ranks on cuda:0 (gloo moves only the
mailbox handles and the timing reductions).  Prints one JSON line (rank 0): loops/s, per-call ms,
collective ms and bytes.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from infiniteexamodels.jl_amd import shard
from infiniteexamodels.jl_amd.model import ExaModel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="farmer", choices=("farmer", "opf", "pandemic", "quadrotor"))
    ap.add_argument("--supports", type=int, default=100_000, help="scenarios (supports) per rank")
    ap.add_argument("--nt", type=int, default=4990, help="pandemic: time supports (+10 extra)")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"))
    ap.add_argument("--same-device", action="store_true")
    ap.add_argument("--allreduce", default="own", choices=("own", "rccl"))
    ap.add_argument("--gpus", type=int, default=0, help="spawn this many ranks (no torchrun needed)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        import bench
        sys.exit(bench.spawn_ranks(args.gpus, script=os.path.abspath(__file__)))

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if args.dist_backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    nccl = args.dist_backend == "nccl"

    from infiniteexamodels.jl_amd import transcribe, workloads
    n_glob = args.supports * world
    t0 = time.perf_counter()
    group = 1
    if args.workload == "farmer":
        gcore = transcribe.exa_core(workloads.farmer(n_glob))
    elif args.workload == "opf":
        gcore = transcribe.exa_core(workloads.opf(n_glob))
    elif args.workload == "pandemic":
        gcore, group = transcribe.exa_core(workloads.pandemic(args.nt, n_glob)), 2
    else:
        gcore = transcribe.exa_core(workloads.quadrotor(n_glob))
    m = ExaModel.sharded(gcore.to_blob(), group, rank, world, device=local_rank)
    del gcore
    info = m.shard_info()
    own = info["own_n"] * ((args.nt + 10) if args.workload == "pandemic" else 1)
    shard.connect_mailboxes(m, dist)
    build_s = time.perf_counter() - t0
    meta = m.meta
    rng = np.random.default_rng(rank)
    x = torch.tensor(np.abs(meta.x0 + 0.1 * rng.standard_normal(meta.nvar)) + 0.05, device=dev)
    y = torch.tensor(rng.standard_normal(meta.ncon), device=dev)
    shared = torch.tensor(np.nonzero(m.shard_var_map()[1] & 2)[0], device=dev)
    buf = torch.zeros(1 + shared.numel(), dtype=torch.float64, device=dev)
    hbuf = torch.zeros(1 + shared.numel(), dtype=torch.float64)          # gloo rehearsal: host copy
    f = torch.zeros(1, dtype=torch.float64, device=dev)
    g = torch.empty(meta.nvar, dtype=torch.float64, device=dev)
    c = torch.empty(meta.ncon, dtype=torch.float64, device=dev)
    jv = torch.empty(meta.nnzj, dtype=torch.float64, device=dev)
    hv = torch.empty(meta.nnzh, dtype=torch.float64, device=dev)

    def collective():
        if args.allreduce == "own":
            m.allreduce_obj_grad(f, g)
        elif nccl:
            shard.allreduce_obj_grad_device(f, g, shared, buf, dist)
        else:
            buf[0:1] = f
            if shared.numel():
                buf[1:] = g[shared]
            hbuf.copy_(buf)
            dist.all_reduce(hbuf)
            buf.copy_(hbuf)
            if shared.numel():
                g[shared] = buf[1:]

    calls = [("halo", lambda: m.halo_exchange(x)), ("obj", lambda: m.obj_device(x, f)), ("grad", lambda: m.grad(x, g)), ("allreduce", collective),
             ("cons", lambda: m.cons(x, c)), ("jac_coord", lambda: m.jac_coord(x, jv)),
             ("hess_coord", lambda: m.hess_coord(x, y, hv, obj_weight=1.0))]

    def loop():
        for _, fn in calls:
            fn()

    def barrier():
        dist.barrier(device_ids=[local_rank]) if nccl else dist.barrier()

    for _ in range(args.warmup):
        loop()
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        loop()
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    ms = {}
    for name, fn in calls:           # per-call device time, after the timed loop
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(args.iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms[name] = e0.elapsed_time(e1) / args.iters
    red = torch.tensor([dt, float(own)], dtype=torch.float64, device=dev if nccl else "cpu")
    tmax = red.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tot = red.clone(); dist.all_reduce(tot)
    torch.cuda.synchronize()
    assert m.comm_status() == 0, "a mailbox wait timed out"
    fglob = float(f.item()) if args.allreduce == "own" else float(buf[0].item())
    if rank == 0:
        dt = float(tmax[0].item())
        print(json.dumps({
            "workload": f"{args.workload}, {int(tot[1].item())} supports over {world} rank(s) ({own} on rank 0), full loop "
                        "obj+grad!+allreduce+cons!+jac_coord!+hess_coord!",
            "n_gpus": world, "dist_backend": args.dist_backend, "allreduce": args.allreduce, "halo_doubles": info["halo_doubles"], "mailbox_kind": m.shard_info()["mailbox_kind"], "same_device": bool(args.same_device), "scaling": "weak",
            "nvar_rank0": meta.nvar, "ncon_rank0": meta.ncon, "nnzj_rank0": meta.nnzj, "nnzh_rank0": meta.nnzh,
            "loop_ms": dt / args.iters * 1e3, "loops_per_s": args.iters / dt,
            "supports_per_s": float(tot[1].item()) * args.iters / dt,
            "collective": {"bytes": 8 * (1 + int(shared.numel())), "shared_entries": int(shared.numel()), "ms": ms["allreduce"]},
            "ms": ms, "build_s": build_s, "obj_global": fglob}), flush=True)
    barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
