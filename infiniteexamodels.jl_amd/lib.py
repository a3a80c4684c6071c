"""ctypes binding of ``libiem_hip.so`` (the C-ABI of ``include/iem.h``).

The library is the product path: if it is missing this module raises — there is no
Python/CPU fallback for evaluation.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libiem_hip.so")
KERNEL_DIR = os.path.join(_HERE, "kernels")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

_lib = None
DEFAULT_STORE_MODE = 2


class IemError(RuntimeError):
    pass


class Meta(C.Structure):
    _fields_ = [("nvar", C.c_int64), ("ncon", C.c_int64), ("npar", C.c_int64), ("nnzj", C.c_int64),
                ("nnzh", C.c_int64), ("n_templates", C.c_int64), ("minimize", C.c_int32),
                ("n_kernels", C.c_int32)]


class TemplateInfo(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("kind", "n_items", "o0", "o1", "o2", "o1step", "o2step")]


class Option(C.Structure):
    _fields_ = [("name", C.c_char_p), ("value", C.c_int64)]


class ShardT(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("group", "rank", "world", "mailbox_kind")] + \
               [(n, C.c_int64) for n in ("n_global", "own_lo", "own_n", "halo", "halo_reach", "halo_doubles", "nvar_global",
                                         "ncon_global", "nnzj_global", "nnzh_global", "nvar", "ncon", "nnzj", "nnzh",
                                         "n_templates", "n_shared")]

    def asdict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class ShardTemplate(C.Structure):
    _fields_ = [("global_index", C.c_int64), ("kind", C.c_int64), ("n_items", C.c_int64), ("klo", C.c_int64 * 3),
                ("dims", C.c_int64 * 3), ("global_dims", C.c_int64 * 3)] + \
               [(n, C.c_int64) for n in ("o0", "o1", "o2", "global_o0", "global_o1", "global_o2", "o1step", "o2step", "items_offset")]

    def asdict(self, items=None):
        """``items``: the concatenated explicit item lists; adds ``ordinals`` = global item ordinal of every local item."""
        import numpy as np
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d = {k: (tuple(int(x) for x in v) if hasattr(v, "__len__") else int(v)) for k, v in d.items()}
        if d["items_offset"] >= 0:
            d["ordinals"] = np.asarray(items[d["items_offset"]:d["items_offset"] + d["n_items"]], dtype=np.int64)
        else:
            k0, k1, k2 = (np.arange(n) for n in d["dims"])
            g0, g1, _ = d["global_dims"]
            d["ordinals"] = ((d["klo"][0] + k0)[None, None, :] + g0 * ((d["klo"][1] + k1)[None, :, None]
                             + g1 * (d["klo"][2] + k2)[:, None, None])).reshape(-1)
        return d


COMM_HANDLE_BYTES = 128


class KktInfo(C.Structure):
    _fields_ = [("S", C.c_int64), ("n", C.c_int64), ("n_border", C.c_int64), ("block_doubles", C.c_int64),
                ("nb", C.c_int32), ("ne", C.c_int32), ("nc", C.c_int32), ("reach", C.c_int32), ("group", C.c_int32), ("phase", C.c_int32),
                ("lanes", C.c_int64), ("hub_ld", C.c_int64), ("hubs", C.c_int32), ("hub_rows", C.c_int32), ("hubs_per_block", C.c_int32), ("reserved_", C.c_int32)]


class KernelInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("kind", C.c_int32), ("jit", C.c_int32), ("grid", C.c_int64 * 3),
                ("lds_bytes", C.c_int64), ("alg_bytes_read", C.c_int64), ("alg_bytes_written", C.c_int64)]


# every symbol include/iem.h declares (tests check the export list against the header)
SYMBOLS = ["iem_create", "iem_create_opts", "iem_create_sharded", "iem_shard_info", "iem_shard_var_map", "iem_shard_template_info",
           "iem_shard_template_items", "iem_shard_blob", "iem_comm_export", "iem_comm_connect", "iem_halo_exchange", "iem_halo_exchange_async", "iem_halo_wait", "iem_halo_reads", "iem_halo_fold", "iem_allreduce_obj_grad", "iem_comm_status",
           "iem_destroy", "iem_meta", "iem_template_info", "iem_kernel_info", "iem_get_host", "iem_set_stream",
           "iem_synchronize", "iem_set_parameter", "iem_obj", "iem_obj_device", "iem_obj_begin", "iem_obj_end", "iem_grad", "iem_cons",
           "iem_jac_coord", "iem_hess_coord", "iem_jac_hess_coord", "iem_eval_trial", "iem_eval_accepted", "iem_eval_all", "iem_jprod", "iem_jtprod", "iem_hprod", "iem_jac_structure", "iem_hess_structure",
           "iem_jac_structure_device", "iem_hess_structure_device", "iem_csr_values", "iem_csr_values32", "iem_csr_spmv", "iem_kkt_chain_factor", "iem_kkt_chain_level", "iem_kkt_hub_level", "iem_kkt_chain_solve", "iem_kkt_chain_solve_lanes", "iem_kkt_source", "iem_kkt_create", "iem_kkt_destroy", "iem_kkt_info", "iem_kkt_layout", "iem_kkt_analyse_blob", "iem_kkt_assemble", "iem_kkt_factor", "iem_kkt_solve", "iem_emit_source", "iem_emit_launch_plan", "iem_blob_hess_structure", "iem_blob_array", "iem_free",
           "iem_set_option", "iem_time_kernels", "iem_tuner_choice", "iem_tune", "iem_last_error", "iem_version"]


def build_library(force: bool = False) -> str:
    """Compile ``libiem_hip.so`` in-tree (host C++; links libamdhip64 + libhiprtc)."""
    src_dir = os.path.join(_HERE, "csrc")
    newest = max(os.path.getmtime(os.path.join(src_dir, f)) for f in os.listdir(src_dir)
                 if f.endswith((".cpp", ".hpp", ".h")) and f != "iem_device_h.inc")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < newest:
        subprocess.check_call(["make", "-C", src_dir, "-s"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IemError(f"{LIB_PATH} is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback for the evaluation path)")
    # One HIP runtime per process: torch (this binding's device-memory provider) ships its own
    # libamdhip64; if libiem_hip.so pulled in the system one FIRST, the two runtimes coexist and
    # the second to initialise sees no device.  Loading torch first makes libiem_hip bind to the
    # runtime torch already loaded (same SONAME).  C / Julia hosts link one runtime and are not
    # affected.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i64, dbl, i32 = C.c_void_p, C.c_int64, C.c_double, C.c_int
    L.iem_last_error.restype = C.c_char_p
    L.iem_version.restype = C.c_char_p
    L.iem_create.argtypes = [C.c_char_p, C.c_size_t, i32, C.POINTER(vp)]
    L.iem_create_opts.argtypes = [C.c_char_p, C.c_size_t, i32, C.POINTER(Option), i32, C.POINTER(vp)]
    L.iem_create_sharded.argtypes = [C.c_char_p, C.c_size_t, i32, i32, i32, i32, C.POINTER(Option), i32, C.POINTER(vp)]
    L.iem_shard_info.argtypes = [vp, C.POINTER(ShardT)]
    L.iem_shard_var_map.argtypes = [vp, vp, vp]
    L.iem_shard_template_info.argtypes = [vp, i64, C.POINTER(ShardTemplate)]
    L.iem_shard_blob.argtypes = [C.c_char_p, C.c_size_t, i32, i32, i32, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(ShardT),
                                 C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.iem_shard_template_items.argtypes = [vp, vp, C.POINTER(C.c_int64)]
    L.iem_comm_export.argtypes = [vp, vp]
    L.iem_comm_connect.argtypes = [vp, vp]
    L.iem_halo_exchange.argtypes = [vp, vp]
    L.iem_halo_exchange_async.argtypes = [vp, vp]
    L.iem_halo_wait.argtypes = [vp]
    L.iem_halo_reads.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.iem_halo_fold.argtypes = [vp, vp]
    L.iem_allreduce_obj_grad.argtypes = [vp, vp, vp]
    L.iem_comm_status.argtypes = [vp, C.POINTER(C.c_int64)]
    L.iem_destroy.argtypes = [vp]
    L.iem_meta.argtypes = [vp, C.POINTER(Meta)]
    L.iem_template_info.argtypes = [vp, i64, C.POINTER(TemplateInfo)]
    L.iem_kernel_info.argtypes = [vp, i32, C.POINTER(KernelInfo)]
    L.iem_get_host.argtypes = [vp, i32, vp]
    L.iem_set_stream.argtypes = [vp, vp]
    L.iem_synchronize.argtypes = [vp]
    L.iem_set_parameter.argtypes = [vp, i64, i64, vp]
    L.iem_obj.argtypes = [vp, vp, C.POINTER(dbl)]
    L.iem_obj_device.argtypes = [vp, vp, vp]
    L.iem_obj_begin.argtypes = [vp, vp]
    L.iem_obj_end.argtypes = [vp, C.POINTER(dbl)]
    L.iem_jac_hess_coord.argtypes = [vp, vp, vp, dbl, vp, vp]
    L.iem_eval_trial.argtypes = [vp, vp, vp, C.POINTER(dbl)]
    L.iem_eval_accepted.argtypes = [vp, vp, vp, dbl, vp, vp, vp]
    L.iem_eval_all.argtypes = [vp, vp, vp, dbl, vp, vp, vp, vp, C.POINTER(dbl)]
    L.iem_grad.argtypes = [vp, vp, vp]
    L.iem_cons.argtypes = [vp, vp, vp]
    L.iem_jac_coord.argtypes = [vp, vp, vp]
    L.iem_hess_coord.argtypes = [vp, vp, vp, dbl, vp]
    L.iem_jprod.argtypes = [vp, vp, vp, vp]
    L.iem_jtprod.argtypes = [vp, vp, vp, vp]
    L.iem_hprod.argtypes = [vp, vp, vp, vp, dbl, vp]
    for f in ("iem_jac_structure", "iem_hess_structure", "iem_jac_structure_device", "iem_hess_structure_device"):
        getattr(L, f).argtypes = [vp, vp, vp, i32]
    L.iem_csr_values.argtypes = [vp, i64, vp, vp, vp, vp]
    L.iem_csr_values32.argtypes = [vp, i64, vp, vp, vp, vp]
    L.iem_csr_spmv.argtypes = [vp, i64, vp, vp, vp, vp, vp, i64, vp]
    L.iem_kkt_chain_factor.argtypes = [vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, dbl]
    L.iem_kkt_chain_level.argtypes = [vp, i64, i64, i32, i32, vp, vp, vp, vp, vp, vp, dbl, i64, i32]
    L.iem_kkt_hub_level.argtypes = [vp, i64, i64, i32, i32, vp, vp, vp, i32, vp, i32, vp, i32, i32, i64, vp, vp, vp, i32]
    L.iem_kkt_chain_solve.argtypes = [vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32]
    L.iem_kkt_chain_solve_lanes.argtypes = [vp, i64, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32]
    L.iem_kkt_source.argtypes = [i32, i32, i32, C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.iem_kkt_create.argtypes = [vp, i32, C.POINTER(vp)]
    L.iem_kkt_destroy.argtypes = [vp]
    L.iem_kkt_info.argtypes = [vp, C.POINTER(KktInfo)]
    L.iem_kkt_layout.argtypes = [vp, vp, vp, vp, vp]
    L.iem_kkt_analyse_blob.argtypes = [vp, C.c_size_t, i32, C.POINTER(KktInfo)] + [C.POINTER(vp)] * 7 + [C.POINTER(i64), C.POINTER(i64)]
    L.iem_kkt_assemble.argtypes = [vp, vp, vp, vp, dbl, dbl]
    L.iem_kkt_factor.argtypes = [vp, vp]
    L.iem_kkt_solve.argtypes = [vp, vp, vp]
    L.iem_emit_source.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)]
    L.iem_emit_source.restype = i32
    L.iem_free.argtypes = [vp]
    L.iem_set_option.argtypes = [C.c_char_p, i64]
    L.iem_time_kernels.argtypes = [vp, vp, vp, vp, vp, i32, C.POINTER(dbl), C.POINTER(dbl)]
    L.iem_tuner_choice.argtypes = [vp, i32, vp, C.POINTER(i32)]
    L.iem_tune.argtypes = [vp, vp, vp, dbl, vp, vp]
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        raise IemError(f"libiem_hip error {rc}: {lib().iem_last_error().decode(errors='replace')}")


def set_option(name: str, value: int):
    check(lib().iem_set_option(name.encode(), int(value)))


# generator knobs and their defaults (csrc/iem_codegen.hpp: struct Options)
OPTION_DEFAULTS = dict(store_mode=2, nt_stores=1, block=0, lds_slots=24, reorder=1, no_fuse=0, hess_merge=0, ablate=0,
                       min_waves=0, fp_contract=0, fuse_zero=1, fuse_groups=1, split_small=64, poll_obj=1, xcd_remap=0, overlap=1, wide_stores=1, obj_wgs=1024, det_shared=1, obj_unroll=1, flat2d=0, flush32=2, autotune=0, autotune_min_blocks=400, pull_scatter=1, fold_colloc=2, fold_max_n=6, det_axis=1, det_scatter=1, det_scatter_max=1 << 28, lazy_loads=2, lazy_min_loads=48, lazy_all_kinds=0, name_tag=0,
                       big_batch_slots=48, big_batch_jac=4000, big_batch_hess=4000, big_xcd=1, big_tile=1024, pair_kernel=1, store_wait=0, comm_timeout_ms=5000,
                       carrier=0, phase_kernels=1, jac_split=1, jac_split_min=0, pair_inter=0, split_shift=0, cons_direct_2d=1)


def option_array(opts: dict):
    """``iem_option_t[]`` for ``iem_create_opts`` (per-handle generator options)."""
    unknown = set(opts) - set(OPTION_DEFAULTS)
    if unknown:
        raise KeyError(f"unknown generator option(s): {sorted(unknown)}")
    arr = (Option * max(1, len(opts)))()
    for i, (k, v) in enumerate(opts.items()):
        arr[i].name = k.encode()
        arr[i].value = int(v)
    return arr, len(opts)


class options:
    """``with options(split_small=0): ...`` — PROCESS DEFAULTS of the generator knobs for the models
    created inside the block, restored on exit.  One model only: ``ExaModel(core, options={...})``
    (``iem_create_opts``), which leaves the defaults alone."""

    def __init__(self, **kw):
        unknown = set(kw) - set(OPTION_DEFAULTS)
        if unknown:
            raise KeyError(f"unknown generator option(s): {sorted(unknown)}")
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            set_option(k, OPTION_DEFAULTS[k])
        return False


def emit_source(blob: bytes):
    """Generated HIP source of a model and its cache key (no device needed)."""
    L = lib()
    L.iem_emit_source.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    p = C.c_void_p()
    key = C.c_uint64()
    check(L.iem_emit_source(blob, len(blob), C.byref(p), C.byref(key)))
    try:
        src = C.string_at(p).decode()
    finally:
        L.iem_free(p)
    return src, int(key.value)


def kkt_analyse_blob(blob: bytes, group: int = 0):
    """Host analysis of the chain KKT solver behind ``iem_kkt_create`` (no device): ``(info dict, blk, loc, rows, cols, dest, seg,
    perm)`` as numpy arrays."""
    import numpy as np
    L = lib()
    info = KktInfo()
    ptrs = [C.c_void_p() for _ in range(7)]
    nd, npm = C.c_int64(), C.c_int64()
    check(L.iem_kkt_analyse_blob(blob, len(blob), int(group), C.byref(info), *[C.byref(p) for p in ptrs], C.byref(nd), C.byref(npm)))
    n = int(info.n)
    shapes = [(n, np.int64), (n, np.int64), (int(info.nc), np.int32), (int(info.nc), np.int32), (nd.value, np.int64), (nd.value + 1, np.uint32), (npm.value, np.uint32)]
    out = []
    try:
        for p, (cnt, dt) in zip(ptrs, shapes):
            out.append(np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int64 if dt == np.int64 else C.c_int32 if dt == np.int32 else C.c_uint32)), shape=(max(cnt, 1),))[:cnt].copy())
    finally:
        for p in ptrs:
            L.iem_free(p)
    return ({k: int(getattr(info, k)) for k, _ in KktInfo._fields_}, *out)


def kkt_source(nb: int, ne: int, nc: int = 12):
    """HIP source of the chain KKT solver's kernels for block size ``nb`` / border size ``ne`` / coupling width ``nc`` and its
    cache key."""
    L = lib()
    p, key = C.c_void_p(), C.c_uint64()
    check(L.iem_kkt_source(int(nb), int(ne), int(nc), C.byref(p), C.byref(key)))
    try:
        src = C.string_at(p).decode()
    finally:
        L.iem_free(p)
    return src, int(key.value)


def precompile_source(src: str, key: int, arch: str = "gfx950", defer: list = None) -> str:
    """Offline-compile a complete HIP source (first line ``// iem-flags: ...``) into the in-tree code-object cache."""
    os.makedirs(KERNEL_DIR, exist_ok=True)
    out = os.path.join(KERNEL_DIR, f"iem_{key:016x}.hsaco")
    if os.path.exists(out):
        return out
    hip = os.path.join(KERNEL_DIR, f"iem_{key:016x}.hip")
    with open(hip, "w") as f:
        f.write(src)
    flags = src.split("\n", 1)[0][len("// iem-flags:"):].split()
    cmd = [os.path.join(ROCM, "bin", "hipcc"), "--genco", f"--offload-arch={arch}", *flags, "-o", out + ".tmp", hip]
    if defer is not None:
        if not any(c[1] == out for c in defer):
            defer.append((cmd, out))
        return out
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


def emit_launch_plan(blob: bytes) -> str:
    L = lib()
    L.iem_emit_launch_plan.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
    p = C.c_void_p()
    check(L.iem_emit_launch_plan(blob, len(blob), C.byref(p)))
    try:
        return C.string_at(p).decode()
    finally:
        L.iem_free(p)


def blob_array(blob: bytes, array_id: int):
    """Model array `array_id` as the library holds it after parsing (float64), including arrays it
    synthesised itself (lattice recovery).  Tooling/tests."""
    import numpy as np
    L = lib()
    L.iem_blob_array.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    p, n = C.c_void_p(), C.c_int64()
    check(L.iem_blob_array(blob, len(blob), int(array_id), C.byref(p), C.byref(n)))
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(max(n.value, 1),))[:n.value].copy()
    finally:
        L.iem_free(p)


def blob_hess_structure(blob: bytes, base: int = 0):
    """Hessian structure of a blob under the current options (host computation, no device)."""
    import numpy as np
    L = lib()
    L.iem_blob_hess_structure.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_int64)]
    r, c, n = C.c_void_p(), C.c_void_p(), C.c_int64()
    check(L.iem_blob_hess_structure(blob, len(blob), base, C.byref(r), C.byref(c), C.byref(n)))
    try:
        rows = np.ctypeslib.as_array(C.cast(r, C.POINTER(C.c_int64)), shape=(max(n.value, 1),))[:n.value].copy()
        cols = np.ctypeslib.as_array(C.cast(c, C.POINTER(C.c_int64)), shape=(max(n.value, 1),))[:n.value].copy()
    finally:
        L.iem_free(r)
        L.iem_free(c)
    return rows, cols


def shard_blob(blob: bytes, group: int, rank: int, world: int):
    """``iem_shard_blob``: rank ``rank``'s shard of a GLOBAL blob, cut in C++ without a device —
    ``(local blob, info dict, var_map, var_flag, [template dicts])``."""
    import numpy as np
    L = lib()
    ob, on, info = C.c_void_p(), C.c_size_t(), ShardT()
    vm, vf, tp, it = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    check(L.iem_shard_blob(blob, len(blob), group, rank, world, C.byref(ob), C.byref(on), C.byref(info),
                           C.byref(vm), C.byref(vf), C.byref(tp), C.byref(it)))
    try:
        local = C.string_at(ob, on.value)
        n = int(info.nvar)
        var_map = np.ctypeslib.as_array(C.cast(vm, C.POINTER(C.c_int64)), shape=(max(n, 1),))[:n].copy()
        var_flag = np.ctypeslib.as_array(C.cast(vf, C.POINTER(C.c_uint8)), shape=(max(n, 1),))[:n].copy()
        tarr = C.cast(tp, C.POINTER(ShardTemplate))
        n_items = sum(int(tarr[i].n_items) for i in range(int(info.n_templates)) if tarr[i].items_offset >= 0)
        items = np.ctypeslib.as_array(C.cast(it, C.POINTER(C.c_int64)), shape=(max(n_items, 1),))[:n_items].copy()
        tpl = [tarr[i].asdict(items) for i in range(int(info.n_templates))]
    finally:
        for p in (ob, vm, vf, tp, it):
            L.iem_free(p)
    return local, info.asdict(), var_map, var_flag, tpl


def precompile(blob: bytes, arch: str = "gfx950", force: bool = False, defer: list = None) -> str:
    """Offline-compile a model's kernels into the in-tree code-object cache
    (``hipcc --genco --offload-arch=gfx950``); returns the ``.hsaco`` path.  With ``defer`` (a
    list) the source is generated now — under the CURRENT generator options — and the hipcc command
    is appended to the list instead of being run: ``run_deferred`` compiles them in parallel."""
    src, key = emit_source(blob)
    os.makedirs(KERNEL_DIR, exist_ok=True)
    out = os.path.join(KERNEL_DIR, f"iem_{key:016x}.hsaco")
    if os.path.exists(out) and not force:
        return out
    hip = os.path.join(KERNEL_DIR, f"iem_{key:016x}.hip")
    with open(hip, "w") as f:
        f.write(src)
    first = src.split("\n", 1)[0]
    assert first.startswith("// iem-flags:"), first
    flags = first[len("// iem-flags:"):].split()
    cmd = [os.path.join(ROCM, "bin", "hipcc"), "--genco", f"--offload-arch={arch}", *flags, "-o", out + ".tmp", hip]
    if defer is not None:
        if not any(c[1] == out for c in defer):
            defer.append((cmd, out))
        return out
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


def run_deferred(jobs: list, workers: int = 0) -> None:
    """Run the hipcc commands collected by ``precompile(..., defer=jobs)`` on a small thread pool."""
    from concurrent.futures import ThreadPoolExecutor

    def one(job):
        cmd, out = job
        subprocess.check_call(cmd)
        os.replace(out + ".tmp", out)
    workers = workers or max(1, min(6, len(os.sched_getaffinity(0)) - 1))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(one, jobs))
