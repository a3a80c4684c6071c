"""C-ABI checks that need no GPU: the library loads, exports every symbol include/iem.h
declares, generates kernels that cross-compile for gfx950, rejects malformed blobs, and
REFUSES to evaluate without a device (no CPU fallback)."""
import ctypes as C
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

import cases
from infiniteexamodels.jl_amd import lib as iemlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "iem.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(iem_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built):
    L = iemlib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(L, name), f"{name} is declared in include/iem.h but not exported"
    assert sorted(iemlib.SYMBOLS) == declared, "lib.py's binding list drifted from include/iem.h"
    assert b"gfx950" in L.iem_version()


def test_create_without_gpu_fails_loudly(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    blob = cases.build_core("quadrotor_5").to_blob()
    h = C.c_void_p()
    rc = iemlib.lib().iem_create(blob, len(blob), 0, C.byref(h))
    assert rc == -5 and not h.value           # IEM_E_NODEVICE
    assert b"no CPU path" in iemlib.lib().iem_last_error()
    from infiniteexamodels.jl_amd.model import ExaModel
    with pytest.raises(iemlib.IemError):
        ExaModel(cases.build_core("quadrotor_5"))


@pytest.mark.parametrize("mutation", ["truncate", "magic", "version", "length", "node_order", "index_range"])
def test_malformed_blobs_are_rejected(mutation, built):
    core = cases.build_core("quadrotor_5")
    blob = bytearray(core.to_blob())
    w = np.frombuffer(blob, dtype=np.int64)
    if mutation == "truncate":
        blob = blob[:len(blob) // 2]
    elif mutation == "magic":
        w[0] ^= 0xFF
    elif mutation == "version":
        w[1] = 99
    elif mutation == "length":
        w[8] += 1
    elif mutation == "node_order":
        tpl_table = 14 + 6 * int(w[6])
        t0 = int(w[tpl_table + 10])          # an ODE template
        n_if, n_ff, n_idx, n_nodes = (int(v) for v in w[t0 + 10:t0 + 14])
        nodes = t0 + 21 + 6 * (n_if + n_ff) + 8 * n_idx
        w[nodes + 4 * (n_nodes - 1) + 1] = n_nodes + 5   # root's child points past the end
    elif mutation == "index_range":
        w[2] = 3                              # nvar far smaller than the indices used
    src = C.c_void_p()
    key = C.c_uint64()
    L = iemlib.lib()
    L.iem_emit_source.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    rc = L.iem_emit_source(bytes(blob), len(blob), C.byref(src), C.byref(key))
    assert rc == -1, mutation
    assert L.iem_last_error()


def test_null_arguments_do_not_crash(built):
    L = iemlib.lib()
    assert L.iem_meta(None, None) == -4
    assert L.iem_obj(None, None, None) == -4
    assert L.iem_set_option(b"no_such_option", 1) == -4
    assert L.iem_destroy(None) == 0
    # per-handle options are validated before anything else happens
    blob = cases.build_core("quadrotor_5").to_blob()
    h = C.c_void_p()
    arr, n = iemlib.option_array({"store_mode": 1})
    arr[0].name = b"no_such_option"
    assert L.iem_create_opts(blob, len(blob), 0, arr, n, C.byref(h)) == -4 and not h.value
    assert b"no_such_option" in L.iem_last_error()
    with pytest.raises(KeyError):
        iemlib.option_array({"nope": 1})


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_generated_kernels_cross_compile_for_gfx950(built, tmp_path):
    """hipcc cross-compiles the emitted source for gfx950 on a machine without a GPU."""
    blob = cases.build_core("test_problem_1").to_blob()
    src, key = iemlib.emit_source(blob)
    assert src.startswith("// iem-flags:")
    hip = tmp_path / "k.hip"
    hip.write_text(src)
    flags = src.split("\n", 1)[0][len("// iem-flags:"):].split()
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--genco", "--offload-arch=gfx950", *flags, "-o",
                           str(tmp_path / "k.hsaco"), str(hip)], stderr=subprocess.DEVNULL)
    assert (tmp_path / "k.hsaco").stat().st_size > 1000
    plan = iemlib.emit_launch_plan(blob)
    assert "kernel iem_jac_" in plan and "kernel iem_hess_" in plan


def test_launch_plan_reports_algorithmic_bytes(lane_fused):
    """Roofline bookkeeping: quadrotor jac reads 6 x-slabs + the stencil array, writes nnzj."""
    from infiniteexamodels.jl_amd import transcribe, workloads
    S = 4096
    plan = iemlib.emit_launch_plan(transcribe.exa_core(workloads.quadrotor(S)).to_blob())
    jac = [l for l in plan.splitlines() if l.startswith("kernel iem_jac_")][0].split()   # (one launch: iem_jac_all = the data body + the computed body)
    rbytes, wbytes = int(jac[jac.index("rbytes") + 1]), int(jac[jac.index("wbytes") + 1])
    assert wbytes == 8 * (62 * S - 18)
    assert 8 * (6 * S + (S - 1)) <= rbytes <= 8 * 7 * S   # 6 x-slabs + the stencil column
