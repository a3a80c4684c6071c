#!/usr/bin/env python3
"""Config 3 of BASELINE.json: pandemic SIR (ESCAPE34/pandemic.jl), full NLPModels evaluation
loop — obj, grad!, cons!, jac_coord!, hess_coord! in solver order — on one MI355X.

  python tools/eval_loop.py [--nt 4990] [--nxi 100] [--iters 100] [--workload pandemic|farmer|quadrotor]

Prints one JSON line with per-call device times (HIP events around each call on the launch
stream) and the achieved HBM rate from the generator's algorithmic byte counts.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import numpy as np
import torch

from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="pandemic")
    ap.add_argument("--nt", type=int, default=4990)     # + 10 extra supports = 5000
    ap.add_argument("--nxi", type=int, default=100)
    ap.add_argument("--supports", type=int, default=100_000)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--opt", action="append", default=[], help="generator knob name=value (repeatable)")
    ap.add_argument("--products", action="store_true", help="also time jprod!/jtprod!/hprod! (event-timed, outside the loop figures)")
    args = ap.parse_args()
    args.options = {k: int(v) for k, v in (kv.split("=") for kv in args.opt)}     # per handle (iem_create_opts)
    print(json.dumps(measure(args)))


def measure(args):
    """One workload through the five-call loop; `args` needs workload, nt, nxi, supports, iters."""
    t0 = time.perf_counter()
    if args.workload == "pandemic":
        im = workloads.pandemic(args.nt, args.nxi)
        desc = f"pandemic SIR, Nt={args.nt + 10} x Nxi={args.nxi} = {(args.nt + 10) * args.nxi} supports"
    elif args.workload == "opf":
        im = workloads.opf(args.supports)
        desc = f"stochastic AC-OPF (3-bus synthetic network), {args.supports} scenarios"
    elif args.workload == "farmer":
        im = workloads.farmer(args.supports)
        desc = f"two-stage farmer, {args.supports} scenarios"
    elif args.workload == "hovercraft":
        im = workloads.hovercraft(args.supports)
        desc = f"hovercraft (examples/hovercraft_example.jl), {args.supports} supports"
    elif args.workload == "three_node":
        im = workloads.three_node_design(args.supports)
        desc = f"3-node design (examples/3node_design.jl), {args.supports} scenarios"
    elif args.workload == "kinetic":
        im = workloads.kinetic_control(args.supports)
        desc = f"kinetic control (examples/kinetic_control.jl), OrthogonalCollocation(4), {args.supports} public supports"
    elif args.workload == "pandemic_oc3":
        im = workloads.pandemic(args.nt, args.nxi, collocation=3)
        desc = f"pandemic SIR, examples/pandemic.jl variant (OrthogonalCollocation(3), u constant over elements), Nt={args.nt + 10} public x Nxi={args.nxi}"
    elif args.workload == "quadrotor_oc3":
        im = workloads.quadrotor(args.supports, collocation=3)
        desc = f"quadrotor ESCAPE34 variant (OrthogonalCollocation(3), piecewise-constant controls), {args.supports} public supports"
    else:
        im = workloads.quadrotor(args.supports)
        desc = f"quadrotor, {args.supports} supports"
    core = transcribe.exa_core(im)
    gm = ExaModel(core, device=0, options=getattr(args, "options", None))
    t_build = time.perf_counter() - t0
    rng = np.random.default_rng(0)
    x0 = gm.meta.x0 + 0.1 * rng.standard_normal(gm.meta.nvar)
    x = torch.tensor(x0 if args.workload in ("opf", "quadrotor", "quadrotor_oc3", "hovercraft") else (gm.meta.x0 if args.workload == "kinetic" else np.abs(x0) + 0.05), device="cuda")
    y = torch.tensor(np.random.default_rng(1).standard_normal(gm.meta.ncon), device="cuda")
    g = torch.empty(gm.meta.nvar, dtype=torch.float64, device="cuda")
    c = torch.empty(gm.meta.ncon, dtype=torch.float64, device="cuda")
    jv = torch.empty(gm.meta.nnzj, dtype=torch.float64, device="cuda")
    hv = torch.empty(gm.meta.nnzh, dtype=torch.float64, device="cuda")
    calls = {
        "obj": lambda: gm.obj(x), "grad": lambda: gm.grad(x, g), "cons": lambda: gm.cons(x, c),
        "jac_coord": lambda: gm.jac_coord(x, jv), "hess_coord": lambda: gm.hess_coord(x, y, hv, obj_weight=1.0),
    }
    timed = dict(calls)
    if getattr(args, "products", False):
        v = torch.tensor(np.random.default_rng(2).standard_normal(gm.meta.nvar), device="cuda")
        Jv = torch.empty(gm.meta.ncon, dtype=torch.float64, device="cuda")
        Jtv = torch.empty(gm.meta.nvar, dtype=torch.float64, device="cuda")
        Hv = torch.empty(gm.meta.nvar, dtype=torch.float64, device="cuda")
        timed.update({"jprod": lambda: gm.jprod(x, v, Jv), "jtprod": lambda: gm.jtprod(x, y, Jtv),
                      "hprod": lambda: gm.hprod(x, y, v, Hv, obj_weight=1.0)})
    for f in timed.values():
        for _ in range(5):
            f()
    torch.cuda.synchronize()
    ms = {}
    for name, f in timed.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms[name] = e0.elapsed_time(e1) / args.iters
    t0 = time.perf_counter()
    for _ in range(args.iters):
        for f in calls.values():
            f()
    torch.cuda.synchronize()
    loop_ms = (time.perf_counter() - t0) / args.iters * 1e3
    # the solver-facing forms of the same loop through prebound C calls (what a compiled host — the Julia ccall shim —
    # pays): (a) the five calls in order, obj returning its scalar before anything else is launched; (b) obj DEFERRED:
    # iem_obj_begin first, iem_obj_end after the last launch; (c) deferred + the one-launch jac/hess pair
    forms = {}
    # (d) ONE LAUNCH PER SOLVER PHASE: iem_eval_trial (obj + cons!, value collected last) + iem_eval_accepted (grad! + jac_coord!
    # + hess_coord!) — the same five results from two launches
    for name, kw in (("five_calls", dict(fused=False, defer_obj=False)), ("obj_deferred", dict(fused=False, defer_obj=True)),
                     ("obj_deferred_fused_pair", dict(fused=True, defer_obj=True)), ("two_phase_launches", dict(phases=True)), ("one_launch", dict(one_launch=True))):
        step = gm.raw_loop(x, y, g, c, jv, hv, obj_weight=1.0, **kw)
        for _ in range(10):
            fval = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            step()
        torch.cuda.synchronize()
        forms[name] = (time.perf_counter() - t0) / args.iters * 1e3
    assert fval == gm.obj(x)
    # the two phases on their own (event-timed blocks, like the per-call figures)
    for name, f in (("eval_trial", lambda: gm.eval_trial(x, c)), ("eval_accepted", lambda: gm.eval_accepted(x, y, g, jv, hv, obj_weight=1.0)),
                    ("eval_all", lambda: gm.eval_all(x, y, c, g, jv, hv, obj_weight=1.0))):
        for _ in range(5):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms[name] = e0.elapsed_time(e1) / args.iters
    # what obj costs IN the loop: the deferred loop against the same loop without any objective call
    L_, h_ = gm._L, gm._h
    px, py, pg, pc, pj, ph = (t.data_ptr() for t in (x, y, g, c, jv, hv))
    def no_obj():
        L_.iem_grad(h_, px, pg); L_.iem_cons(h_, px, pc); L_.iem_jac_coord(h_, px, pj); L_.iem_hess_coord(h_, px, py, 1.0, ph)
    for _ in range(10):
        no_obj()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        no_obj()
    torch.cuda.synchronize()
    forms["four_calls_without_obj"] = (time.perf_counter() - t0) / args.iters * 1e3
    forms["obj_in_loop_us"] = {"blocking": (forms["five_calls"] - forms["four_calls_without_obj"]) * 1e3,
                               "deferred": (forms["obj_deferred"] - forms["four_calls_without_obj"]) * 1e3}
    # the same loop captured in a HIP graph (obj kept on the device)
    fdev = torch.zeros(1, dtype=torch.float64, device="cuda")

    def dev_loop():
        gm.obj_device(x, fdev); gm.grad(x, g); gm.cons(x, c); gm.jac_coord(x, jv); gm.hess_coord(x, y, hv, obj_weight=1.0)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dev_loop()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        dev_loop()
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        graph.replay()
    torch.cuda.synchronize()
    graph_ms = (time.perf_counter() - t0) / args.iters * 1e3
    kinds = {"obj": "obj", "grad": "grad", "cons": "cons", "jac_coord": "jac", "hess_coord": "hess"}
    if getattr(args, "products", False):
        kinds.update({"jprod": "jprod", "jtprod": "jtprod", "hprod": "hprod"})
    bytes_ = {k: sum(kk["alg_bytes_read"] + kk["alg_bytes_written"] for kk in gm.kernels() if kk["kind"] == v)
              for k, v in kinds.items()}
    out = {
        "workload": desc, "nvar": gm.meta.nvar, "ncon": gm.meta.ncon, "nnzj": gm.meta.nnzj, "nnzh": gm.meta.nnzh,
        "n_kernels": gm.meta.n_kernels, "build_s": t_build, "loop_ms": loop_ms, "loops_per_s": 1e3 / loop_ms,
        "graph_loop_ms": graph_ms, "graph_loops_per_s": 1e3 / graph_ms, "loop_forms_ms": forms,
        "ms": ms, "alg_bytes": bytes_,
        "GBps": {k: (bytes_[k] / (ms[k] * 1e-3) / 1e9 if ms[k] > 0 else None) for k in ms if k in bytes_},
        "loop_alg_bytes": sum(bytes_[k] for k in ("obj", "grad", "cons", "jac_coord", "hess_coord")),
        "loop_frac_of_8TBps": {k: sum(bytes_[q] for q in ("obj", "grad", "cons", "jac_coord", "hess_coord")) / (v * 1e-3) / 8e12
                               for k, v in forms.items() if isinstance(v, float) and k != "four_calls_without_obj"},
        "phase_kernels": sorted(kk["name"] for kk in gm.kernels() if kk["kind"] in ("trial", "accepted", "point")),
    }
    gm.close()
    return out


if __name__ == "__main__":
    main()
