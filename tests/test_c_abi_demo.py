"""The C-ABI from plain C (examples/c_abi_demo.c): compiled with gcc against include/iem.h,
run as a child process on the GPU box, compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

import cases
from pyoracle import OracleModel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "infiniteexamodels.jl_amd")


def compile_demo(out):
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L" + PKG, "-liem_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-o", out])


def test_c_demo_compiles_against_the_header(built, tmp_path):
    compile_demo(str(tmp_path / "c_abi_demo"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["quadrotor_100", "opf_7"])
def test_c_demo_matches_oracle(name, built, tmp_path):
    exe = str(tmp_path / "c_abi_demo")
    compile_demo(exe)
    core = cases.build_core(name)
    blob = core.to_blob()
    om = OracleModel(blob)
    x, y = cases.eval_point_for(name, om, seed=2)
    (tmp_path / "m.blob").write_bytes(blob)
    x.astype("<f8").tofile(tmp_path / "x.f64")
    y.astype("<f8").tofile(tmp_path / "y.f64")
    r = subprocess.run([exe, str(tmp_path / "m.blob"), str(tmp_path / "x.f64"), str(tmp_path / "y.f64")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = {l.split()[0]: l.split()[1:] for l in r.stdout.splitlines()}
    assert [int(v) for v in got["meta"][:4]] == [om.nvar, om.ncon, om.nnzj, om.nnzh]
    ref = {"obj": np.array([om.obj(x)]), "grad": om.grad(x), "cons": om.cons(x), "jac": om.jac_coord(x), "hess": om.hess_coord(x, y, 1.0)}
    for k, v in ref.items():
        n, s, q = int(got[k][0]), float(got[k][1]), float(got[k][2])
        assert n == v.size
        scale = max(1.0, float(np.abs(v).sum()))
        assert abs(s - v.sum()) <= 1e-10 * scale, (k, s, v.sum())
        assert abs(q - (v * v).sum()) <= 1e-10 * max(1.0, float((v * v).sum())), k
    jr, jc = om.jac_structure(base=1)
    assert [int(v) for v in got["jac_structure"]] == [om.nnzj, int(jr.sum()), int(jc.sum())]
    # the deferred objective and the one-launch jac + hess pair give the very same numbers as the five plain calls
    assert got["obj2"] == got["obj"] and got["jac2"] == got["jac"] and got["hess2"] == got["hess"]
    # ... and so does one launch per solver phase (iem_eval_trial, iem_eval_accepted)
    assert got["obj3"] == got["obj"] and all(got[k + "3"] == got[k] for k in ("cons", "grad", "jac", "hess"))
    # the KKT solve from C (iem_kkt_*): K = [H + 0.01 I, J'; J, -1e-6 I] at this point, right-hand side (grad; cons)
    import scipy.sparse as sp
    from scipy.sparse.linalg import spsolve
    hr, hc = om.hess_structure(base=0)
    jr0, jc0 = om.jac_structure(base=0)
    hv, jv = om.hess_coord(x, y, 1.0), om.jac_coord(x)
    n, m = om.nvar, om.ncon
    off = hr != hc
    H = sp.coo_matrix((np.concatenate([hv, hv[off]]), (np.concatenate([hr, hc[off]]), np.concatenate([hc, hr[off]]))), shape=(n, n))
    J = sp.coo_matrix((jv, (jr0, jc0)), shape=(m, n))
    K = sp.bmat([[H + 1e-2 * sp.identity(n), J.T], [J, -1e-6 * sp.identity(m)]]).tocsc()
    want = spsolve(K, np.concatenate([om.grad(x), om.cons(x)]))
    neg = int((np.linalg.eigvalsh(K.toarray()) < 0).sum())     # (a random point: the Hessian block need not be positive definite)
    assert [int(v) for v in got["kkt_inertia"]] == [n + m - neg, neg, 0]
    cnt, s, q = int(got["kkt_solution"][0]), float(got["kkt_solution"][1]), float(got["kkt_solution"][2])
    assert cnt == n + m
    assert abs(s - want.sum()) <= 1e-6 * max(1.0, float(np.abs(want).sum()))
    assert abs(q - (want * want).sum()) <= 1e-6 * max(1.0, float((want * want).sum()))
