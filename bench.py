#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json: jac_coord!+hess_coord! evaluation pairs
per second on the quadrotor transcription (examples/quadrotor.jl) at 10^6 supports,
with the achieved fraction of the gfx950 HBM roofline and a CPU baseline timed in the
same run.

  python bench.py --gpus N --steps K --warmup W [--supports S] [--scaling strong|weak]

A "step" is one jac_coord! + one hess_coord! over the resident model (inputs already in HBM): the metric's own call pair,
iem_jac_coord + iem_hess_coord — what an NLPModels solver makes (ext/InfiniteExaModelsIpopt.jl:48-49,
ext/InfiniteExaModelsMadNLP.jl:49-50).  The C-ABI's one-launch form of the pair (iem_jac_hess_coord, identical bytes) is
reported beside it as "fused_pair" (`--fused` makes IT the timed step; then "separate_calls" is the one beside).  N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); the time
axis is sharded into contiguous support blocks (halo of one support for the
finite-difference rows); jac_coord!/hess_coord! need no collective, so ranks only meet
at the barriers that bracket the timed region.

Launch: under torchrun (RANK/WORLD_SIZE in the environment) the script is one rank; started
plainly with --gpus N > 1 it spawns its N ranks itself — fresh child processes, created before
this process makes any GPU call — and rank 0 prints the one JSON line.

Scaling: STRONG by default (SURVEY.md §7 "Strong scaling to >= 6x"): the headline problem —
10^6 supports in total — is cut into N shards of 10^6/N supports (+ halo); `value` = evaluation
pairs per second of that fixed problem, so value(N)/value(1) is the speed-up.

N > 1, what the timed step is: every solver iteration moves x, so the stencil neighbours x_k[a_r - 1] must cross
the shard boundary before the rows that read them are evaluated.  The timed step is therefore
`iem_halo_exchange_async(x); jac_coord!; hess_coord!` — the exchange INSIDE the timed region (key "halo_in_timed_loop").
The library launches nothing for it: the exchange rides on the step's first launch as one extra leading workgroup, because
that kernel's loads cannot touch a halo entry (csrc/iem_api.cpp: halo_plan; a call that can gets the stand-alone exchange
kernel in front of it); the communication-free pair is reported beside it ("pair_no_halo").
Before the ranks wire their mailboxes in-process, the same wiring + one checked exchange + one checked all-reduce
run in CHILD processes (key "comm": peer-mapped memory is the one part that the one-GPU box can only rehearse with all
ranks on one device — whatever it does on a real xGMI node must end in an entry of the line, never in a lost line).  If
the children fail, a rank cannot wire its mailbox, or a mailbox wait times out during the warm-up, the timed step takes
the torch.distributed fallback of the exchange instead (shard.ShardComm: a send / recv of the halo doubles, RCCL on GPUs) —
still INSIDE the timed step; "comm_path" says which ("own" / "rccl"; "none" only if the fallback failed too, and then the
value is not a multi-GPU step).  `--force-rccl` takes the fallback deliberately.  `--weak` adds the weak form of the same run under "weak".

Inputs: x and y are resident in HBM and re-used every step (a solver's iterate); `roofline.frac_cold_inputs` is the same
pair with K = 4 distinct (x, y) sets cycled per call (> 256 MiB in total: nothing of a call's inputs can sit in the
Infinity Cache from the call before) — measured outside the timed region.

Generator options go through iem_create_opts (per HANDLE; `--opt name=value`), never the process defaults.  The staging
batch of jac_coord! / hess_coord! is a function of the grid size (config.lds_slots reports what each kernel got).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
PMC_PROFILE = "pmc_quadrotor_1e6.json"   # committed rocprofv3 --pmc passes of this command (tools/profile_gpu.sh; static, not live)


def csrc_fingerprint() -> str:
    """sha256 over the library's sources (csrc/*.cpp, *.hpp, *.h): the PMC profile records it, so a profile taken from
    another csrc tree is visible in the line (`roofline.traffic_stale`).  (No git on the GPU box: content, not commit.)"""
    import hashlib
    d = os.path.join(ROOT, "infiniteexamodels.jl_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".cpp", ".hpp", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def eval_point(nvar, ncon, x0, S_local, seed=0):
    """x = x0 + 0.1 N(0,1) (seed 0), |x8| clipped to 1.2; y ~ N(0,1) (seed 1)."""
    x = x0 + 0.1 * np.random.default_rng(seed).standard_normal(nvar)
    x[7 * S_local:8 * S_local] = np.clip(x[7 * S_local:8 * S_local], -1.2, 1.2)
    y = np.random.default_rng(seed + 1).standard_normal(ncon)
    return x, y


def cpu_baseline(sample_supports: int, seconds: float = 12.0):
    """CPU baseline timed on this box's host cores on a bounded sample (the same quadrotor model
    at `sample_supports` supports), reported in the metric's unit (pairs/s normalised to 1e6
    supports).  `value`: the reference algorithm's structure — one loop per template, zero-fill
    then accumulate, no fusion — COMPILED for the host (oracle/cpu_compiled_baseline.py; the
    closest stand-in for ExaModels' type-specialised CPU loops, which cannot run here), one
    thread.  Also reported: the same with all cores (OpenMP), and the generic tree interpreter
    oracle/iem_oracle.c that the parity tests use."""
    from infiniteexamodels.jl_amd import transcribe, workloads
    from pyoracle import OracleModel
    import cpu_compiled_baseline as cb
    core = transcribe.exa_core(workloads.quadrotor(sample_supports))
    om = OracleModel(core.to_blob())
    x, y = eval_point(om.nvar, om.ncon, om.x0, sample_supports)
    scale = sample_supports / 1e6
    ncores = min(len(os.sched_getaffinity(0)), 16)   # the box's CPU share for one GPU
    r1, jv, hv = cb.time_pairs(core, x, y, om.nnzj, om.nnzh, seconds=seconds / 3, threads=1)
    rall, _, _ = cb.time_pairs(core, x, y, om.nnzj, om.nnzh, seconds=seconds / 4, threads=ncores)
    om.set_threads(1)
    t0 = time.perf_counter()
    jo = om.jac_coord(x)
    ho = om.hess_coord(x, y, 1.0)
    interp = scale / (time.perf_counter() - t0)
    agree = float(max(np.abs(jv - jo).max(), np.abs(hv - ho).max()))
    return {
        "value": r1 * scale, "unit": "jac+hess pairs/s (1e6-support quadrotor equivalent)", "cores": 1,
        "kind": "port",
        "sample": f"quadrotor at {sample_supports} supports; per-template compiled host loops (reference launch "
                  f"structure: one loop per template, fill!(0) then +=, no cross-template fusion), g++ -O3 "
                  f"-march=native -ffp-contract=off, scaled by supports/1e6",
        "value_all_cores": rall * scale, "all_cores": ncores,
        "interpreter_value": interp, "interpreter": "oracle/iem_oracle.c (generic tree interpreter), 1 thread",
        "max_abs_diff_vs_oracle": agree,
    }


def live_traffic(args):
    """HBM bytes per launch of the jac / hess / pair kernels from the PMC counters, collected NOW on this box: two rocprofv3
    `--pmc` passes (FETCH_SIZE, WRITE_SIZE: separate passes, as MI355X_MICROARCH.md prescribes; FETCH_SIZE tallies 128-byte
    requests at 64 B on gfx950 -> x 2; both in KiB) of a child process that builds the same model and launches each kernel a
    few times.  Run BEFORE this process touches the GPU; any failure yields None (the line then falls back to the committed
    profile).  Returns ({kernel name: bytes per launch}, note)."""
    import csv, glob, shutil, subprocess, tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="iem_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
               "--supports", str(args.supports), "--store-mode", str(args.store_mode), "--nt", str(args.nt), "--fma", str(args.fma), "--hess-layout", args.hess_layout] + \
              [a for kv in args.opt for a in ("--opt", kv)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", IEM_PMC_CHILD="1"), capture_output=True, text=True, timeout=240)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {r.returncode}): {(r.stderr or r.stdout).strip().splitlines()[-1][:200] if (r.stderr or r.stdout).strip() else ''}"
            acc = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Kernel_Name"].startswith("iem_") and row["Counter_Name"] == counter:
                        acc.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            if not acc:
                return None, f"no {counter} rows in the rocprofv3 output"
            for k, v in acc.items():
                out.setdefault(k, {})[counter] = sum(v) / len(v)
        except Exception as e:      # noqa: BLE001 — the bench line must not depend on the profiler
            return None, f"{counter} pass: {e}"[:300]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    res = {k: (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 for k, c in out.items() if "FETCH_SIZE" in c and "WRITE_SIZE" in c}
    return res, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of a child of THIS run: (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, mean over its launches"


def choose_comm_path(children_ok: bool, wired: bool, warmup_status: int = 0, rccl_ok: bool = True):
    """Which exchange the N > 1 TIMED step contains — never silently none:
      'own'   the library's mailbox kernels (iem_halo_exchange_async riding on the step's first launch)
      'rccl'  torch.distributed send / recv of the halo doubles (shard.ShardComm's fallback; backend nccl = RCCL over xGMI):
              when the child-process check failed, a rank could not export / map a mailbox, or a mailbox wait timed out
              during the warm-up
      'none'  only when the fallback itself failed: the line then carries comm.path = 'none' and the reason
    Returns (path, reason)."""
    if children_ok and wired and warmup_status == 0:
        return "own", ""
    why = ("the child-process check of the mailbox path failed" if not children_ok else
           "a rank could not export / map a mailbox" if not wired else f"a mailbox wait timed out during the warm-up (status {warmup_status})")
    if rccl_ok:
        return "rccl", why
    return "none", why + "; the torch.distributed fallback failed too"


def spawn_ranks(n: int, script: str = None) -> int:
    """Plain `python bench.py --gpus N` (no RANK in the environment): start the N ranks as fresh
    child processes — this parent has not touched the GPU and never does — and wait for them.
    Rank 0 inherits stdout and prints the JSON line.  A rank that dies takes the others with it
    (exact PIDs), so a failure cannot hang at a barrier."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = rc or code
                for q in live:
                    q.terminate()
    return rc


def comm_section(gm, step, torch, dist, dev, red_dev, barrier, rank, world, iters, connect=True):
    """Multi-GPU data path around the pair, through the library's own mailboxes (iem_comm_export / _connect,
    iem_halo_exchange, iem_allreduce_obj_grad).  Every stage is agreed on by all ranks (MIN over an ok flag)
    before the next one starts, and the device-side waits are bounded, so a failure yields an "error" entry
    in the line instead of a hang.  The halo is checked exactly: x[i] = f(global index of i) on owned entries,
    NaN on the halo copies — after the exchange every halo copy must hold f(its global index)."""
    def agreed(ok: bool) -> bool:
        t = torch.tensor([1.0 if ok else 0.0], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    out, err, ok = {}, "", True
    if connect:     # (False: the ranks of this process are wired already)
        try:
            mine = gm.comm_export()
        except Exception as e:     # noqa: BLE001 — reported in the line
            mine, err = b"", f"export: {e}"
        handles = [None] * world
        dist.all_gather_object(handles, mine)
        ok = all(isinstance(h, (bytes, bytearray)) and len(h) == len(handles[0]) and len(h) > 0 for h in handles)
        if ok:
            try:
                gm.comm_connect(b"".join(handles))
            except Exception as e:     # noqa: BLE001
                ok, err = False, f"connect: {e}"
    if not agreed(ok):
        return {"error": err or "a peer could not export / map a mailbox"}
    try:
        info = gm.shard_info()
        out["mailbox_kind"], out["halo_doubles"] = info["mailbox_kind"], info["halo_doubles"]
        vmap, vflag = gm.shard_var_map()
        f = lambda g: ((g * 2654435761) % 1000003).astype(np.float64) / 1000003.0 + 0.25
        xh = f(vmap)
        halo = (vflag & 4) != 0
        xh[halo] = np.nan
        xc = torch.tensor(xh, device=dev)
        fo = torch.tensor([float(rank + 1)], device=dev, dtype=torch.float64)
        barrier()
        gm.halo_exchange(xc)
        gm.allreduce_obj_grad(fo, None)
        torch.cuda.synchronize()
        got = xc.cpu().numpy()
        halo_ok = bool(np.array_equal(got, f(vmap)))
        sum_ok = float(fo.item()) == world * (world + 1) / 2
        out["halo_exact"], out["allreduce_exact"] = halo_ok, sum_ok
        ok = halo_ok and sum_ok and gm.comm_status() == 0
    except Exception as e:     # noqa: BLE001
        ok, err = False, f"check: {e}"
    if not agreed(ok):
        out["error"] = err or "halo / all-reduce check failed on a rank"
        return out
    try:
        def timed(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(5):
                fn()
            barrier(); torch.cuda.synchronize()
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            t = torch.tensor([e0.elapsed_time(e1) / iters * 1e3], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        out["halo_exchange_us"] = timed(lambda: gm.halo_exchange(xc))
        out["allreduce_obj_us"] = timed(lambda: gm.allreduce_obj_grad(fo, None))
        us = timed(lambda: (gm.halo_exchange(xc), step()))
        out["pair_behind_blocking_halo"] = {"ms_per_step": us / 1e3, "note": "stream-ordered iem_halo_exchange in front of every pair (round-2 form; the "
                                            "headline step uses the asynchronous exchange); device time, max over ranks"}
        ok = gm.comm_status() == 0
    except Exception as e:     # noqa: BLE001
        ok, err = False, f"timing: {e}"
    if not agreed(ok):
        out["error"] = err or "a mailbox wait timed out"
    return out


def comm_isolated(args, dist, rank, world, local_rank, barrier):
    """Run comm_section in CHILD processes (one per rank, on this rank's GPU, gloo for their host rendezvous) while
    the parents idle: peer-mapped memory is the one part of this script that has only been rehearsed with all ranks
    on a single GPU, and whatever it does on a real xGMI node — an error code, a time-out, a fault — must end in a
    "comm": {"error": …} entry, never in a lost headline line."""
    import socket
    import subprocess
    port = [0]
    if rank == 0:
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port[0] = sk.getsockname()[1]; sk.close()
    dist.broadcast_object_list(port, src=0)
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local_rank), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port[0]), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE", "GROUP_RANK", "ROLE_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--comm-child", "--gpus", str(world), "--dist-backend", "gloo", "--supports", str(args.supports),
           "--scaling", args.scaling, "--steps", str(min(args.steps, 200)), "--warmup", "3", "--store-mode", str(args.store_mode),
           "--hess-layout", args.hess_layout, "--no-cold"] + (["--same-device"] if args.same_device else []) + (["--fused"] if args.fused else []) + [a for kv in args.opt for a in ("--opt", kv)]
    out, rc = "", -1
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            out, errtxt = p.communicate(timeout=120)
            rc = p.returncode
        except subprocess.TimeoutExpired:
            p.kill()
            out, errtxt = p.communicate()
            rc = -9
    except Exception as e:     # noqa: BLE001
        errtxt = str(e)
    barrier()
    if rank != 0:
        return None
    for ln in out.splitlines():
        if ln.startswith("COMM "):
            c = json.loads(ln[5:])
            c["isolation"] = "child processes, one per rank"
            return c
    return {"error": f"comm child exited {rc}: {errtxt.strip().splitlines()[-1] if errtxt.strip() else 'no output'}"[:400]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--supports", type=int, default=1_000_000, help="supports in total (strong) / per rank (weak)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong")
    ap.add_argument("--weak", action="store_true", help="N > 1: also measure the weak form (every rank a full --supports shard of an N-times "
                    "longer horizon: N x by construction, and every rank transcribes the N-times larger model first) — off by default")
    ap.add_argument("--no-weak", action="store_true", help=argparse.SUPPRESS)   # (the default now; kept so older command lines parse)
    ap.add_argument("--store-mode", type=int, default=2)
    ap.add_argument("--nt", type=int, default=1)
    ap.add_argument("--fma", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="generator knob name=value (repeatable)")
    ap.add_argument("--hess-layout", choices=("exa", "merged"), default="exa",
                    help="'merged' is an opt-in extension (not the reference's COO layout; not the headline metric)")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="gloo + --same-device rehearses the N>1 path with several ranks on ONE GPU")
    ap.add_argument("--same-device", action="store_true", help="every rank uses cuda:0 (rehearsal only)")
    ap.add_argument("--emulate-shard", default="", help="R/N: time shard R of an N-way sharded run on this GPU")
    ap.add_argument("--graph", action="store_true", help="replay the jac+hess pair from a captured HIP graph instead of eager launches")
    ap.add_argument("--no-comm", action="store_true", help="N > 1: skip the (untimed) check of the halo exchange / objective all-reduce")
    ap.add_argument("--no-comm-check", action="store_true", help="N > 1: skip only the child-process check; the exchange stays in the timed step")
    ap.add_argument("--comm-child", action="store_true", help=argparse.SUPPRESS)   # internal: run ONLY the comm section (see comm_isolated)
    ap.add_argument("--fused", action="store_true", help="the timed step is iem_jac_hess_coord (one launch) instead of the metric's own call pair "
                    "iem_jac_coord + iem_hess_coord (two launches, the default); the other form is always reported beside it")
    ap.add_argument("--separate", action="store_true", help=argparse.SUPPRESS)   # (the default; kept so older command lines still parse)
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-input measurement (K rotating x / y sets)")
    ap.add_argument("--rotate-inputs", type=int, default=0, metavar="K",
                    help="cold-input measurement (roofline.frac_cold_inputs): K distinct (x, y) sets cycled per call; 0 = as many as "
                         "it takes to exceed the 256-MiB Infinity Cache, at least 4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the secondary measurement of the ESCAPE34 (collocation) variant of the model")
    ap.add_argument("--cpu-sample", type=int, default=100_000)
    ap.add_argument("--no-live-traffic", action="store_true", help="do not collect roofline.traffic with rocprofv3 --pmc child passes (then: the committed profile, flagged static)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # internal: build the model, launch jac / hess / pair a few times, exit
    ap.add_argument("--rehearse-comm", default="", help=argparse.SUPPRESS)   # with --rehearse-launch: simulate one branch of choose_comm_path
    ap.add_argument("--force-rccl", action="store_true", help="N > 1: take the torch.distributed fallback of the halo exchange even when the mailboxes work")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="no GPU, no evaluation, NOT a measurement: spawn / rendezvous / reduce only (CPU test of the N>1 launch path)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    live, live_note = None, None
    if (args.gpus == 1 and "RANK" not in os.environ and not args.no_live_traffic and not args.pmc_child and not args.emulate_shard and not args.rehearse_launch
            and not os.environ.get("IEM_PMC_CHILD") and not any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
            and "rocprof" not in os.environ.get("LD_PRELOAD", "")):
        live, live_note = live_traffic(args)     # before this process touches the GPU: one process on the card at a time

    import torch
    import torch.distributed as dist
    from infiniteexamodels.jl_amd import lib as iemlib
    from infiniteexamodels.jl_amd import shard, transcribe, workloads
    from infiniteexamodels.jl_amd.model import ExaModel

    if args.pmc_child:
        # what rocprofv3 --pmc profiles for `roofline.traffic`: the same model, each kernel launched a few times
        torch.cuda.set_device(0)
        hopts = {"store_mode": args.store_mode, "nt_stores": args.nt, "fp_contract": args.fma}
        for kv in args.opt:
            k, v = kv.split("=")
            hopts[k] = int(v)
        core = transcribe.exa_core(workloads.quadrotor(args.supports))
        gm = ExaModel(core, device=0, hess_layout=args.hess_layout, options=hopts)
        x, y = eval_point(gm.meta.nvar, gm.meta.ncon, gm.meta.x0, gm.meta.nvar // 22)
        xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
        jac = torch.empty(gm.meta.nnzj, dtype=torch.float64, device="cuda")
        hess = torch.empty(gm.meta.nnzh, dtype=torch.float64, device="cuda")
        for _ in range(4):
            gm.jac_coord(xd, jac); gm.hess_coord(xd, yd, hess, obj_weight=1.0); gm.jac_hess_coord(xd, yd, jac, hess, obj_weight=1.0)
        torch.cuda.synchronize()
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    if args.same_device:
        local_rank = 0
    if args.rehearse_launch:
        # launch-path rehearsal for the CPU suite: the same spawn, environment, rendezvous and
        # max-over-ranks reduction as a real run, nothing evaluated, nothing timed
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        out = {"rehearsal": True, "n_gpus": world, "max_rank_seen": int(t.item())}
        if args.rehearse_comm:
            # every branch of the comm-path selection with SIMULATED outcomes, agreed on by all ranks like the real run does
            sim = {"own": (True, True, 0, True), "children_failed": (False, True, 0, True), "connect_failed": (True, False, 0, True),
                   "warmup_timeout": (True, True, 4 if rank == world - 1 else 0, True), "rccl_failed": (True, False, 0, False)}[args.rehearse_comm]
            st = torch.tensor([float(sim[2])], dtype=torch.float64)
            dist.all_reduce(st, op=dist.ReduceOp.MAX)          # one rank's time-out is everybody's
            path, why = choose_comm_path(sim[0], sim[1], int(st.item()), sim[3])
            out["comm"] = {"path": path, "why": why, "world_size": dist.get_world_size()}
        if rank == 0:
            print(json.dumps(out), flush=True)
        dist.destroy_process_group()
        return
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torchrun (RANK set) the process group is always created — also at world 1, so the
    # RCCL path of this script can be rehearsed on a one-GPU box
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    red_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")

    def barrier():
        if use_dist:
            if args.dist_backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
    if not os.path.exists(iemlib.LIB_PATH):   # the prebuilt in-tree library is the normal case; never let N ranks run make at once
        if rank == 0:
            iemlib.build_library()
        barrier()
    # generator options of THIS run's handles (iem_create_opts): nothing process-global
    hopts = {"store_mode": args.store_mode, "nt_stores": args.nt, "fp_contract": args.fma, "comm_timeout_ms": 1000}
    for kv in args.opt:
        k, v = kv.split("=")
        hopts[k] = int(v)

    # N > 1: the comm path is first exercised in child processes (see the module docstring); only when every child came
    # back clean do the ranks of THIS process wire their mailboxes and put the exchange into the timed step
    comm = None
    want_halo = world > 1 and use_dist and not args.no_comm and not args.emulate_shard and not args.comm_child
    if want_halo and not args.no_comm_check and not (args.same_device and 2 * world > 6):
        comm = comm_isolated(args, dist, rank, world, local_rank, barrier)
        ok = [comm is None or "error" not in comm]
        if use_dist:
            dist.broadcast_object_list(ok, src=0)
        children_ok = bool(ok[0])      # a failed check does NOT drop the exchange from the timed step: the torch.distributed fallback takes it
    else:
        children_ok = True
    if args.force_rccl:
        children_ok = False

    def measure(scaling: str, steps: int, warmup: int, with_halo: bool):
        """Build this rank's shard for `scaling`, run `warmup` untimed + exactly `steps` timed steps between barriers,
        return (max-over-ranks seconds, supports all ranks own, state).  A step is one jac_coord! + one hess_coord!
        (`--fused`: the one-launch form), preceded on a wired multi-GPU run by the asynchronous halo exchange of x."""
        if scaling == "weak":
            S_global = args.supports * world
        else:
            S_global = args.supports
        # N > 1: every rank holds the GLOBAL blob — what a host that knows nothing about sharding has —
        # and iem_create_sharded cuts this rank's window of the time axis (group 1) inside the library
        er, ew = rank, world
        if args.emulate_shard:     # build shard R of N on this one GPU (no communication): "R/N"
            er, ew = (int(v) for v in args.emulate_shard.split("/"))
            S_global = args.supports * ew if scaling == "weak" else args.supports
        core = transcribe.exa_core(workloads.quadrotor(S_global))
        blob = core.to_blob()
        if ew == 1:
            gm = ExaModel(core, device=local_rank, blob=blob, hess_layout=args.hess_layout, options=hopts)
            S_local = S_global
        else:
            del core
            gm = ExaModel.sharded(blob, 1, er, ew, device=local_rank, hess_layout=args.hess_layout, options=hopts)
            S_local = gm.shard_info()["own_n"]
        del blob
        x, y = eval_point(gm.meta.nvar, gm.meta.ncon, gm.meta.x0, gm.meta.nvar // 22, seed=rank * 2)
        xd, yd = torch.tensor(x, device=dev), torch.tensor(y, device=dev)
        jac = torch.empty(gm.meta.nnzj, dtype=torch.float64, device=dev)
        hess = torch.empty(gm.meta.nnzh, dtype=torch.float64, device=dev)
        halo_state = {"in_loop": False, "path": "none"}
        sc = None
        if with_halo:
            # the library's own mailboxes when every rank can wire them; otherwise (child check failed, export / connect
            # refused, --force-rccl) the torch.distributed fallback of shard.ShardComm — the exchange is INSIDE the timed step
            # either way (choose_comm_path)
            sc = shard.ShardComm(gm, dist, force_fallback=not children_ok)
            path, why = choose_comm_path(children_ok, sc.kind == "own")
            if not children_ok and args.force_rccl:
                why = "--force-rccl"
            halo_state.update(kind=sc.kind, why=why or sc.why, path=path, in_loop=True,
                              mailbox_kind=gm.shard_info()["mailbox_kind"] if sc.kind == "own" else 0)

        def make_step(path):
            """one step through the C-ABI; argument checks and the stream lookup of the Python wrapper are done once
            (ExaModel.raw_pair) — at 8 GPUs a step is 25 us of device time"""
            if path == "rccl":
                pair = gm.raw_pair(xd, yd, jac, hess, obj_weight=1.0, fused=args.fused, halo=False)

                def step():
                    sc.halo_exchange(xd)      # dist.batch_isend_irecv of the halo doubles (RCCL on GPUs)
                    pair()
                return step
            return gm.raw_pair(xd, yd, jac, hess, obj_weight=1.0, fused=args.fused, halo=path == "own")
        step = make_step(halo_state["path"])
        gm._tuned = gm.tune(xd, yd, jac, hess, obj_weight=1.0)   # no-op unless --opt autotune=1

        if args.graph:
            def eager():      # the wrappers follow torch's CURRENT stream, which the capture needs
                if halo_state["path"] == "own":
                    gm.halo_exchange_async(xd)
                if args.fused:
                    gm.jac_hess_coord(xd, yd, jac, hess, obj_weight=1.0)
                else:
                    gm.jac_coord(xd, jac)
                    gm.hess_coord(xd, yd, hess, obj_weight=1.0)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                eager()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                eager()
            step = graph.replay
        barrier()      # ranks build their shards at different speeds: start the first exchanges together
        for _ in range(warmup):
            step()
        if halo_state["path"] == "own":
            # the warm-up steps carried real exchanges over this machine's links: if ANY rank saw a mailbox wait time out
            # (bounded: comm_timeout_ms = 1 s on these handles), every rank takes the torch.distributed fallback instead and
            # the line says so — a broken link must cost seconds, not steps x time-out
            st = gm.comm_status()
            flag = torch.tensor([float(st)], device=red_dev, dtype=torch.float64)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if flag.item() != 0:
                try:
                    gm.synchronize()      # reports and clears the recorded time-out
                except Exception:         # noqa: BLE001
                    pass
                path, why = choose_comm_path(True, True, int(flag.item()))
                sc = shard.ShardComm(gm, dist, force_fallback=True)
                halo_state.update(path=path, why=why, kind=sc.kind)
                step = make_step(path)
        if halo_state["path"] == "rccl":
            # the fallback's own rehearsal: a few steps, agreed on by all ranks; if IT fails the line says comm.path = none
            ok = True
            try:
                for _ in range(max(2, warmup)):
                    step()
                torch.cuda.synchronize()
            except Exception as e:        # noqa: BLE001
                ok = False
                halo_state["rccl_error"] = str(e)[:300]
            flag = torch.tensor([0.0 if ok else 1.0], device=red_dev, dtype=torch.float64)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if flag.item() != 0:
                path, why = choose_comm_path(False, False, 0, rccl_ok=False)
                halo_state.update(path=path, why=halo_state.get("why", "") + "; " + why, in_loop=False)
                step = make_step("none")
        barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # HIP events over the timed region, on the launch stream
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            step()
        ev1.record()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        halo_state["loop_event_ms"] = ev0.elapsed_time(ev1) / steps
        if halo_state["path"] == "own":
            halo_state["status"] = gm.comm_status()     # 0: every exchange of the timed loop completed
        if use_dist:
            every = [None] * world
            dist.all_gather_object(every, dt / steps * 1e3)
            halo_state["rank_ms_per_step"] = {"min": min(every), "max": max(every), "all": every}
            t = torch.tensor([dt], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            sl = torch.tensor([float(S_local)], device=red_dev, dtype=torch.float64)
            dist.all_reduce(sl)
            supports_total = float(sl.item())
        else:
            supports_total = float(S_local)
        return dt, supports_total, (gm, S_local, xd, yd, jac, hess, step, halo_state)

    dt, supports_total, (gm, S_local, xd, yd, jac, hess, step, halo_state) = measure(args.scaling, args.steps, args.warmup, want_halo)

    if args.comm_child:
        ccomm = comm_section(gm, step, torch, dist, dev, red_dev, barrier, rank, world, min(args.steps, 200))
        if rank == 0:
            print("COMM " + json.dumps(ccomm), flush=True)
        barrier()
        dist.destroy_process_group()
        return

    def timed_loop(fn, n):
        """max-over-ranks seconds of n calls of fn between barriers (secondary measurements, outside the headline region)"""
        for _ in range(5):
            fn()
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(); barrier()
        d = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([d], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        return d

    n2 = min(args.steps, 200)
    secondary = {}
    # the other launch form of the same pair, and (N > 1) the same step without the exchange
    other = gm.raw_pair(xd, yd, jac, hess, obj_weight=1.0, fused=not args.fused, halo=halo_state["path"] == "own")
    secondary["fused_pair" if not args.fused else "separate_calls"] = {"ms_per_step": timed_loop(other, n2) / n2 * 1e3}
    if halo_state["in_loop"]:
        nohalo = gm.raw_pair(xd, yd, jac, hess, obj_weight=1.0, fused=args.fused, halo=False)
        secondary["pair_no_halo"] = {"ms_per_step": timed_loop(nohalo, n2) / n2 * 1e3}

    # distribution of single steps (median, p10/p90), HIP events on the
    # launch stream, outside the timed region
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n2 + 1)]
    ev[0].record()
    for e in ev[1:]:
        step()
        e.record()
    torch.cuda.synchronize()
    pair_ms = np.array([a.elapsed_time(b) for a, b in zip(ev[:-1], ev[1:])])

    has_pair = any(k["kind"] == "pair" for k in gm.kernels())

    def block_ms(fn, n=100):
        """average launch duration of ONE kernel: n launches back to back between ONE pair of HIP events on the launch stream
        (an event pair around every launch adds ~3 us to each; rocprofv3 --kernel-trace --stats of this command reports the same average)"""
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n
    ms_jac_blk = block_ms(lambda: gm.jac_coord(xd, jac))
    ms_hess_blk = block_ms(lambda: gm.hess_coord(xd, yd, hess, obj_weight=1.0))
    pair_kernel_ms = block_ms(lambda: gm.jac_hess_coord(xd, yd, jac, hess, obj_weight=1.0)) if has_pair else None
    # the same pair with COLD inputs: K distinct (x, y) sets, > 256 MiB in total, cycled per call — per-kernel events
    cold = None
    if world == 1 and not args.no_cold and not args.graph:
        K = max(4, int(np.ceil(300e6 / (8.0 * (gm.meta.nvar + gm.meta.ncon)))))
        K = max(2, args.rotate_inputs) if args.rotate_inputs else min(K, 64)
        xs = [xd] + [xd + 1e-3 * (i + 1) for i in range(K - 1)]
        ys = [yd] + [yd * (1.0 + 1e-3 * (i + 1)) for i in range(K - 1)]
        nc = 20 * K if gm.meta.nvar < 5_000_000 else 4 * K

        def rot_ms(fn):      # nc calls, inputs cycled per call, between ONE event pair
            for i in range(K):
                fn(i)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(nc):
                fn(i % K)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / nc
        cold = {"sets": K, "input_bytes_total": int(8 * K * (gm.meta.nvar + gm.meta.ncon)),
                "jac_ms": rot_ms(lambda i: gm.jac_coord(xs[i], jac)),
                "hess_ms": rot_ms(lambda i: gm.hess_coord(xs[i], ys[i], hess, obj_weight=1.0)),
                "two_calls_ms": rot_ms(lambda i: (gm.jac_coord(xs[i], jac), gm.hess_coord(xs[i], ys[i], hess, obj_weight=1.0)))}
        if has_pair:
            cold["pair_kernel_ms"] = rot_ms(lambda i: gm.jac_hess_coord(xs[i], ys[i], jac, hess, obj_weight=1.0))
        del xs, ys

    reads_all = None
    if world > 1 and use_dist and halo_state["path"] == "own":
        reads_all = [None] * world       # rank 0 has no left neighbour: what the stencil makes a rank wait for shows on rank 1
        dist.all_gather_object(reads_all, {k: v[0] for k, v in gm.halo_reads().items() if k in ("cons", "jac", "hess", "pair")})
    line = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = supports_total / 1e6 * args.steps / dt
        # roofline of the dominant kernel: its algorithmic bytes / its average launch duration, the latter from blocks of
        # back-to-back launches between ONE pair of HIP events on the launch stream (block_ms) — never per-launch event pairs
        ms_jac, ms_hess = ms_jac_blk, ms_hess_blk
        ks = {k["kind"]: k for k in gm.kernels() if k["kind"] in ("jac", "hess") and k["grid"][0] > 1}
        pair_alg = sum(k["alg_bytes_read"] + k["alg_bytes_written"] for k in ks.values())     # two calls: each reads its own inputs
        fused_kd = dict([k for k in gm.kernels() if k["kind"] == "pair"][0]) if has_pair else None
        fused_alg = (fused_kd["alg_bytes_read"] + fused_kd["alg_bytes_written"]) if has_pair else None   # ONE launch: inputs both bodies load counted once
        if args.fused and has_pair:
            # the timed step is ONE kernel (both calls' workgroups behind one dispatcher): it is the dominant kernel
            dom = "pair"
            kd = fused_kd
            alg, ms_dom = fused_alg, pair_kernel_ms
        else:
            dom = "hess" if ms_hess >= ms_jac else "jac"
            kd = dict(ks[dom])
            if gm._tuned.get(dom, -1) == 1:     # the tuner chose the handle's second code object for this buffer: its kernels carry a tag
                kd["name"] += "_b48"
            alg = kd["alg_bytes_read"] + kd["alg_bytes_written"]
            ms_dom = ms_hess if dom == "hess" else ms_jac
        achieved = alg / (ms_dom * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters cannot be collected inside this process (they need rocprofv3
        # passes of their own): read from the committed profile of the same command and size (tools/profile_gpu.sh →
        # profiles/), which records the fingerprint of the csrc tree it was taken from — `traffic_stale` says whether
        # that is THIS tree.
        traffic, traffic_src, stale, traffic_static = None, None, None, True
        prof = os.path.join(ROOT, "profiles", PMC_PROFILE)
        if live and kd["name"] in live:
            traffic, traffic_src, stale, traffic_static = live[kd["name"]], live_note, False, False
        elif os.path.exists(prof) and S_local == 1_000_000 and world == 1:
            try:
                pj = json.load(open(prof))
                traffic = pj["pmc"][kd["name"]]["hbm_bytes_per_launch"]
                stale = pj.get("csrc_fingerprint") != csrc_fingerprint()
                traffic_src = (f"profiles/{PMC_PROFILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                               f"csrc fingerprint {pj.get('csrc_fingerprint', 'n/a')}, commit {pj.get('commit', 'n/a')})")
            except Exception:
                traffic = None
        par = f"support-sharded x{world}, {args.scaling} scaling" + (f", shard {args.emulate_shard} emulated on one GPU" if args.emulate_shard else "")
        step_txt = ("iem_jac_hess_coord (one launch)" if args.fused else "iem_jac_coord + iem_hess_coord") + \
                   (" behind iem_halo_exchange_async(x)" if halo_state["path"] == "own" else
                    " behind a torch.distributed send/recv of the halo doubles (shard.ShardComm fallback)" if halo_state["path"] == "rccl" else "")
        roof = {"bound": "hbm", "kernel": kd["name"], "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_static": traffic_static,
                "traffic_all_kernels": live, "traffic_note": None if live else live_note,
                "traffic_stale": stale, "traffic_source": traffic_src, "csrc_fingerprint": csrc_fingerprint(),
                "alg_bytes": alg, "kernel_ms": ms_dom, "kernel_ms_from": "100 back-to-back launches between one pair of HIP events on the launch stream",
                "jac_ms": ms_jac, "hess_ms": ms_hess,
                "jac_frac": (ks["jac"]["alg_bytes_read"] + ks["jac"]["alg_bytes_written"]) / (ms_jac * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "hess_frac": (ks["hess"]["alg_bytes_read"] + ks["hess"]["alg_bytes_written"]) / (ms_hess * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "pair_alg_bytes": fused_alg if (args.fused and has_pair) else pair_alg,
                # the step's own fraction, from the HIP events that bracket the TIMED loop (all K steps between one event pair)
                "step_event_ms": halo_state.get("loop_event_ms"),
                "pair_frac": (fused_alg if (args.fused and has_pair) else pair_alg) / (halo_state["loop_event_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if pair_kernel_ms is not None:
            roof["pair_kernel_ms"] = pair_kernel_ms
            roof["pair_kernel_alg_bytes"] = fused_alg
            roof["pair_kernel_frac"] = fused_alg / (pair_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if cold:
            cms = cold.get("pair_kernel_ms") if dom == "pair" else cold["hess_ms"] if dom == "hess" else cold["jac_ms"]
            roof["frac_cold_inputs"] = alg / (cms * 1e-3) / 1e9 / HBM_PEAK_GBS
            roof["pair_frac_cold_inputs"] = pair_alg / (cold["two_calls_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            roof["cold_inputs"] = cold
        line = {
            "metric": "jac_coord!+hess_coord! evals/sec, quadrotor 1e6 supports; % HBM roofline",
            "value": value, "unit": "jac+hess pairs/s (1e6-support quadrotor equivalent)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"quadrotor (examples/quadrotor.jl), backward FD, {int(supports_total)} supports total, "
                                   f"{S_local} per GPU, jac_coord!+hess_coord! only, seed 0/1 inputs resident in HBM",
                       "nvar": gm.meta.nvar, "ncon": gm.meta.ncon, "nnzj": gm.meta.nnzj, "nnzh": gm.meta.nnzh,
                       "store_mode": args.store_mode, "hess_layout": args.hess_layout, "parallelism": par,
                       "step": step_txt, "options": "per handle (iem_create_opts)",
                       "launch": "hipGraph replay" if args.graph else "eager",
                       "lds_bytes": {k: v["lds_bytes"] for k, v in ks.items()},
                       "store_batch_tuner": {**gm._tuned, "meaning": "-1 tuner off (default: the staging batch is a function of kind and grid size)"},
                       "kernels_from": "hiprtc at run time (code-object cache miss)" if any(k["jit"] for k in gm.kernels()) else "in-tree code-object cache"},
            "roofline": roof,
        }
        line["launch"] = {"world_size": dist.get_world_size() if use_dist else 1, "device_count": torch.cuda.device_count(), "process_group": bool(use_dist),
                          "rank_ms_per_step": halo_state.get("rank_ms_per_step")}
        line["pair_ms"] = {"median": float(np.median(pair_ms)), "p10": float(np.percentile(pair_ms, 10)),
                           "p90": float(np.percentile(pair_ms, 90)), "n": int(pair_ms.size)}
        for k, v in secondary.items():
            v["value"] = supports_total / 1e6 / (v["ms_per_step"] * 1e-3)
            line[k] = v
        if world > 1:
            line["halo_in_timed_loop"] = bool(halo_state["in_loop"])
            line["comm_path"] = {"path": halo_state["path"], "why": halo_state.get("why", ""), "backend": args.dist_backend,
                                 "meaning": "own = the library's mailbox kernels; rccl = torch.distributed send/recv inside the timed step; none = no exchange could be made (the value then is NOT a multi-GPU step)"}
            if halo_state.get("rccl_error"):
                line["comm_path"]["rccl_error"] = halo_state["rccl_error"]
            if halo_state["path"] == "own":
                line["halo"] = {"exchange": "iem_halo_exchange_async (deferred; rides on the first launch of the step as one extra workgroup)", "mailbox_kind": halo_state.get("mailbox_kind"),
                                "status_after_timed_loop": halo_state.get("status"), "reads_halo_rank1": reads_all[1] if reads_all else None,
                                "meaning": "reads_halo: which calls of rank 1 need the exchange to be complete (their kernels can load a halo entry of x); the others can carry it"}
            elif not args.no_comm and not args.emulate_shard:
                line["comm_fallback"] = (comm or {}).get("error") or halo_state.get("why") or "not attempted"
            if comm is not None:
                line["comm"] = comm
    # one-GPU rehearsal with many ranks: parents + children would exceed the box's per-GPU process limit — the checked
    # comm section runs in-process there
    if world > 1 and use_dist and not args.no_comm and not args.no_comm_check and not args.emulate_shard and args.same_device and 2 * world > 6:
        ccomm = comm_section(gm, step, torch, dist, dev, red_dev, barrier, rank, world, n2, connect=halo_state["path"] != "own")
        if rank == 0:
            line["comm"] = ccomm
    # secondary measurement, outside the headline timed region: the weak form of the same run
    # (every rank a full `--supports` shard of an N-times longer horizon)
    if world > 1 and args.scaling == "strong" and args.weak and not args.emulate_shard:
        del gm, xd, yd, jac, hess, step, other
        secondary.clear()
        torch.cuda.empty_cache()
        wdt, wsupports, state = measure("weak", max(10, args.steps // 4), max(3, args.warmup // 4), False)
        if rank == 0:
            wsteps = max(10, args.steps // 4)
            line["weak"] = {"value": wsupports / 1e6 * wsteps / wdt, "ms_per_step": wdt / wsteps * 1e3, "steps": wsteps,
                            "supports_total": int(wsupports), "supports_per_gpu": int(state[1]),
                            "note": "every rank owns a full shard; communication-free pair, so this is Nx by construction"}
        del state
    # secondary measurement, outside the headline timed region: the reference's own benchmark VARIANT of the same model
    # (ESCAPE34/quadrotor.jl:13-14,73 — OrthogonalCollocation(3), controls constant over each element) at the same total
    # number of supports (S / 2 public ones + one collocation node per element), same step, same roofline arithmetic
    if world == 1 and not args.emulate_shard and not args.no_variants and not args.graph:
        del gm, xd, yd, jac, hess, step, other
        torch.cuda.empty_cache()
        vcore = transcribe.exa_core(workloads.quadrotor(max(2, args.supports // 2), collocation=3))
        vm = ExaModel(vcore, device=local_rank, hess_layout=args.hess_layout, options=dict(hopts, name_tag=34))   # (kernel names end in _b34: a profile of this command keeps the two models apart)
        vx, vy = eval_point(vm.meta.nvar, vm.meta.ncon, vm.meta.x0, vm.meta.nvar // 22, seed=0)
        vxd, vyd = torch.tensor(vx, device=dev), torch.tensor(vy, device=dev)
        vj = torch.empty(vm.meta.nnzj, dtype=torch.float64, device=dev)
        vh = torch.empty(vm.meta.nnzh, dtype=torch.float64, device=dev)
        vstep = vm.raw_pair(vxd, vyd, vj, vh, obj_weight=1.0, fused=args.fused, halo=False)
        nv = min(args.steps, 100)
        vdt = timed_loop(vstep, nv)
        vjac, vhess = vm.time_kernels(vxd, vyd, vj, vh, iters=50)
        vks = {k["kind"]: k for k in vm.kernels() if k["kind"] in ("jac", "hess") and k["grid"][0] > 1}
        valg = sum(k["alg_bytes_read"] + k["alg_bytes_written"] for k in vks.values())
        vsup = vm.meta.nvar // 22
        line["escape34_variant"] = {
            "workload": f"quadrotor (ESCAPE34/quadrotor.jl: OrthogonalCollocation(3), controls constant over elements), {vsup} supports in total "
                        f"({max(2, args.supports // 2)} public), jac_coord!+hess_coord! only",
            "nvar": vm.meta.nvar, "ncon": vm.meta.ncon, "nnzj": vm.meta.nnzj, "nnzh": vm.meta.nnzh,
            "ms_per_step": vdt / nv * 1e3, "value": vsup / 1e6 * nv / vdt, "unit": "jac+hess pairs/s (per 1e6 supports in total)",
            "jac_ms": vjac, "hess_ms": vhess, "pair_alg_bytes": valg, "pair_frac": valg / ((vjac + vhess) * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "step_frac": valg / (vdt / nv) / 1e9 / HBM_PEAK_GBS}
        vm.close()
        del vm, vxd, vyd, vj, vh, vstep
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        print(json.dumps(line), flush=True)
    if use_dist:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
