"""Parity against vectors dumped from a REAL session of the reference
(infiniteexamodels.jl_amd/julia/dump_reference.jl).  None can be produced in this image (no
julia); the tests collect whatever directories exist under tests/golden/reference/ and are
skipped when there are none — they are the hook that turns "parity unpinned" into "pinned" the
day a maintainer drops a dump in.  The reader/comparator itself is exercised on a dump written
by this repository's own oracle in the dump format (test_dump_format_roundtrip)."""
import glob
import os

import numpy as np
import pytest

import cases
from pyoracle import OracleModel

HERE = os.path.dirname(os.path.abspath(__file__))
DUMPS = sorted(d for d in glob.glob(os.path.join(HERE, "golden", "reference", "*")) if os.path.isdir(d))
RTOL = 1e-10


def read_dump(d):
    def arr(name, dt):
        return np.fromfile(os.path.join(d, name), dtype=dt)
    out = dict(blob=open(os.path.join(d, "model.blob"), "rb").read(), x=arr("x.f64", "<f8"), y=arr("y.f64", "<f8"))
    for k in ("obj", "grad", "cons", "jac_vals", "hess_vals"):
        out[k] = arr(k + ".f64", "<f8")
    for k in ("jac_rows", "jac_cols", "hess_rows", "hess_cols"):
        out[k] = arr(k + ".i64", "<i8")
    meta = dict(line.split(None, 1) for line in open(os.path.join(d, "meta.txt")).read().splitlines() if line.strip())
    out["obj_weight"] = float(meta.get("obj_weight", 1.0))
    out["base"] = int(meta.get("index_base", 1))
    return out


def write_dump(d, blob, x, y, ow=1.0):
    """The dump format, written from the oracle (format self-test only)."""
    om = OracleModel(blob)
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "model.blob"), "wb").write(blob)
    jr, jc = om.jac_structure(base=1)
    hr, hc = om.hess_structure(base=1)
    for name, v in (("x.f64", x), ("y.f64", y), ("obj.f64", np.array([om.obj(x)])), ("grad.f64", om.grad(x)),
                    ("cons.f64", om.cons(x)), ("jac_vals.f64", om.jac_coord(x)), ("hess_vals.f64", om.hess_coord(x, y, ow)),
                    ("jac_rows.i64", jr), ("jac_cols.i64", jc), ("hess_rows.i64", hr), ("hess_cols.i64", hc)):
        np.asarray(v, dtype="<f8" if name.endswith("f64") else "<i8").tofile(os.path.join(d, name))
    open(os.path.join(d, "meta.txt"), "w").write(f"nvar {om.nvar}\nncon {om.ncon}\nobj_weight {ow}\nindex_base 1\n")


def close(a, b):
    np.testing.assert_allclose(a, b, rtol=RTOL, atol=RTOL * max(1.0, float(np.abs(b).max()) if b.size else 1.0))


def compare(ref, obj, grad, cons, jr, jc, jv, hr, hc, hv):
    assert np.array_equal(jr, ref["jac_rows"]) and np.array_equal(jc, ref["jac_cols"])
    assert np.array_equal(hr, ref["hess_rows"]) and np.array_equal(hc, ref["hess_cols"])
    close(np.array([obj]), ref["obj"]); close(grad, ref["grad"]); close(cons, ref["cons"])
    close(jv, ref["jac_vals"]); close(hv, ref["hess_vals"])


def run_oracle(ref):
    om = OracleModel(ref["blob"])
    x, y = ref["x"], ref["y"]
    jr, jc = om.jac_structure(base=ref["base"])
    hr, hc = om.hess_structure(base=ref["base"])
    compare(ref, om.obj(x), om.grad(x), om.cons(x), jr, jc, om.jac_coord(x), hr, hc, om.hess_coord(x, y, ref["obj_weight"]))


def test_dump_format_roundtrip(tmp_path):
    core = cases.build_core("quadrotor_5")
    blob = core.to_blob()
    om = OracleModel(blob)
    rng = np.random.default_rng(5)
    d = str(tmp_path / "quadrotor_5")
    write_dump(d, blob, om.x0 + 0.1 * rng.standard_normal(om.nvar), rng.standard_normal(om.ncon), 0.7)
    run_oracle(read_dump(d))


@pytest.mark.skipif(not DUMPS, reason="no dumps from a real session of the reference under tests/golden/reference/")
@pytest.mark.parametrize("d", DUMPS or ["-"], ids=[os.path.basename(d) for d in DUMPS] or ["none"])
def test_oracle_matches_reference_dump(d):
    run_oracle(read_dump(d))


def run_hip(ref):
    import torch
    from infiniteexamodels.jl_amd.model import ExaModel
    m = ExaModel.from_blob(ref["blob"], device=0)
    x = torch.tensor(ref["x"], device="cuda"); y = torch.tensor(ref["y"], device="cuda")
    jr, jc = m.jac_structure(base=ref["base"])
    hr, hc = m.hess_structure(base=ref["base"])
    compare(ref, m.obj(x), m.grad(x).cpu().numpy(), m.cons(x).cpu().numpy(), np.asarray(jr), np.asarray(jc),
            m.jac_coord(x).cpu().numpy(), np.asarray(hr), np.asarray(hc),
            m.hess_coord(x, y, obj_weight=ref["obj_weight"]).cpu().numpy())


@pytest.mark.gpu
def test_hip_dump_format_roundtrip(tmp_path):
    core = cases.build_core("pandemic_20x3")
    blob = core.to_blob()
    om = OracleModel(blob)
    rng = np.random.default_rng(6)
    d = str(tmp_path / "pandemic")
    write_dump(d, blob, om.x0 + 0.1 * rng.standard_normal(om.nvar), rng.standard_normal(om.ncon), 1.3)
    run_hip(read_dump(d))


@pytest.mark.gpu
@pytest.mark.skipif(not DUMPS, reason="no dumps from a real session of the reference under tests/golden/reference/")
@pytest.mark.parametrize("d", DUMPS or ["-"], ids=[os.path.basename(d) for d in DUMPS] or ["none"])
def test_hip_matches_reference_dump(d):
    run_hip(read_dump(d))
