#!/usr/bin/env python3
"""Does the ALLOCATION API decide the placement mode of jac_coord!'s output buffer (the same kernel takes
0.083 ms into some buffers and 0.097 ms into others; a plain fill moves the same way; hipMalloc re-draws do not help)?
For each way of obtaining the buffer — hipMalloc, the virtual-memory API (one physical handle for the whole buffer; one
handle per 2 MiB / 64 MiB), hipExtMallocWithFlags (fine-grained / uncached), hipMallocManaged — allocate it several times in one
process and time jac_coord! (through the C-ABI, raw pointers) and hipMemsetAsync into it.

  python tools/probes/alloc_api_probe.py [--supports 1000000] [--draws 6]
"""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from infiniteexamodels.jl_amd import transcribe, workloads
from infiniteexamodels.jl_amd.model import ExaModel

ap = argparse.ArgumentParser()
ap.add_argument("--supports", type=int, default=1_000_000)
ap.add_argument("--draws", type=int, default=6)
ap.add_argument("--iters", type=int, default=40)
args = ap.parse_args()

hip = C.CDLL("libamdhip64.so")
vp, sz = C.c_void_p, C.c_size_t


class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class AllocFlags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]


class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", Loc), ("win32HandleMetaData", vp), ("allocFlags", AllocFlags)]


class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]


def chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hipError {rc}")


gm = ExaModel(transcribe.exa_core(workloads.quadrotor(args.supports)), device=0)
x = torch.tensor(gm.meta.x0 + 0.1 * np.random.default_rng(0).standard_normal(gm.meta.nvar), device="cuda")
nbytes = gm.meta.nnzj * 8
stream = torch.cuda.current_stream().cuda_stream
gm._sync_stream()


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters


def measure(ptr):
    jac = timed(lambda: chk(gm._L.iem_jac_coord(gm._h, C.c_void_p(x.data_ptr()), C.c_void_p(ptr)), "iem_jac_coord"))
    fill = timed(lambda: chk(hip.hipMemsetAsync(C.c_void_p(ptr), 0, sz(nbytes), C.c_void_p(stream)), "hipMemsetAsync"))
    return jac, fill


def vmm(chunk=None):
    prop = Prop(); prop.type = 1; prop.location = Loc(1, 0)      # pinned device memory on device 0
    gran = sz()
    chk(hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 1), "granularity")
    g = gran.value
    size = (nbytes + g - 1) // g * g
    ptr = vp()
    chk(hip.hipMemAddressReserve(C.byref(ptr), sz(size), sz(0), vp(0), C.c_ulonglong(0)), "reserve")
    handles = []
    step = size if chunk is None else max(g, chunk // g * g)
    off = 0
    while off < size:
        n = min(step, size - off)
        h = vp()
        chk(hip.hipMemCreate(C.byref(h), sz(n), C.byref(prop), C.c_ulonglong(0)), "hipMemCreate")
        chk(hip.hipMemMap(vp(ptr.value + off), sz(n), sz(0), h, C.c_ulonglong(0)), "hipMemMap")
        handles.append((h, off, n))
        off += n
    acc = Access(Loc(1, 0), 3)
    chk(hip.hipMemSetAccess(ptr, sz(size), C.byref(acc), sz(1)), "hipMemSetAccess")

    def free():
        for h, off, n in handles:
            hip.hipMemUnmap(vp(ptr.value + off), sz(n)); hip.hipMemRelease(h)
        hip.hipMemAddressFree(ptr, sz(size))
    return ptr.value, free, g


def plain():
    p = vp()
    chk(hip.hipMalloc(C.byref(p), sz(nbytes)), "hipMalloc")
    return p.value, lambda: hip.hipFree(p), None


def ext(flags):
    def f():
        p = vp()
        chk(hip.hipExtMallocWithFlags(C.byref(p), sz(nbytes), C.c_uint(flags)), "hipExtMallocWithFlags")
        return p.value, lambda: hip.hipFree(p), None
    return f


def managed():
    p = vp()
    chk(hip.hipMallocManaged(C.byref(p), sz(nbytes), C.c_uint(1)), "hipMallocManaged")
    hip.hipMemPrefetchAsync(p, sz(nbytes), C.c_int(0), C.c_void_p(stream))
    torch.cuda.synchronize()
    return p.value, lambda: hip.hipFree(p), None


ways = {"hipMalloc": plain, "vmm_one_handle": lambda: vmm(None), "vmm_2MiB_handles": lambda: vmm(2 << 20), "vmm_64MiB_handles": lambda: vmm(64 << 20),
        "hipExtMalloc_finegrained": ext(0x1), "hipExtMalloc_uncached": ext(0x3), "hipMallocManaged": managed}
out = {"supports": args.supports, "bytes": nbytes, "draws": args.draws, "ways": {}}
for name, fn in ways.items():
    rows, keep = [], []
    try:
        for d in range(args.draws):
            ptr, free, g = fn()
            jac, fill = measure(ptr)
            rows.append({"jac_ms": jac, "fill_ms": fill, "fill_GBps": nbytes / fill / 1e6})
            keep.append(free)          # held until the way is done: the next draw gets other pages
        out["ways"][name] = rows
        print(name, "granularity", g, " jac_ms", [round(r["jac_ms"], 4) for r in rows], " fill GB/s", [round(r["fill_GBps"]) for r in rows], flush=True)
    except Exception as e:     # noqa: BLE001
        out["ways"][name] = {"error": str(e)}
        print(name, "FAILED", e, flush=True)
    for f in keep:
        f()
print(json.dumps(out))
