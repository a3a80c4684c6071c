"""The blob as a FOREIGN producer would write it: the Julia writer in
julia/MI355XBackend.jl knows nothing about support grids or item boxes — it sees a flat Vector of
NamedTuples per template and writes every item field as an explicit column (arithmetic
progressions as affine fields), one dimension, a grid hint only for unit-step index columns.
Here every model of the suite is re-serialised that way and must evaluate identically to the
native encoding: oracle vs oracle bit for bit, generated kernels (emulated) vs oracle."""
import copy

import numpy as np
import pytest

import cases
from emu import EmulatedModel
from infiniteexamodels.jl_amd.items import Field, Items
from pyoracle import OracleModel


def flatten_template(t):
    """Same template over a 1-D explicit iterator, fields as the Julia writer would emit them."""
    n = len(t.items)
    first_ap = [None]

    def conv(f, is_int):
        col = f.values(t.items.dims)
        if is_int:
            col = np.ascontiguousarray(col, dtype=np.int64)
            step = int(col[1] - col[0]) if n > 1 else 0
            if n == 0 or np.array_equal(col, col[0] + step * np.arange(n)):
                if step == 1 and first_ap[0] is None:
                    first_ap[0] = int(col[0]) - 1
                return Field("int", "affine", int(col[0]) if n else 0, (step,))
            return Field("int", "gather", 0, (1,), col)
        return Field("float", "gather", 0, (1,), np.ascontiguousarray(col, dtype=np.float64))

    u = copy.copy(t)
    u.ifields = [conv(f, True) for f in t.ifields]
    u.ffields = [conv(f, False) for f in t.ffields]
    grid = None
    if first_ap[0] is not None:
        grid = ((4096,), (first_ap[0],))       # MI355XBackend.jl: gid = 4097, origin = first index - 1
    elif n == 1:
        grid = ((), ())
    u.items = Items((max(n, 0),) if n else (0,), {}, grid=grid)
    return u


def foreign_blob(core):
    c = copy.copy(core)
    c.templates = [flatten_template(t) for t in core.templates]
    return c, c.to_blob()


NAMES = [n for n in cases.small_cases() if n not in ("quadrotor_1000", "quadrotor_oc3_700", "pandemic_300x7", "farmer_1000", "opf_600")]


@pytest.mark.parametrize("name", NAMES)
def test_foreign_encoding_evaluates_identically(name, grid_mode):
    core = cases.build_core(name)
    native = OracleModel(core.to_blob())
    fcore, fblob = foreign_blob(core)
    om = OracleModel(fblob)
    assert (om.nvar, om.ncon, om.nnzj, om.nnzh) == (native.nvar, native.ncon, native.nnzj, native.nnzh)
    for a, b in zip(om.jac_structure() + om.hess_structure(), native.jac_structure() + native.hess_structure()):
        assert np.array_equal(a, b)
    x, y = cases.eval_point_for(name, native, seed=4)
    assert om.obj(x) == native.obj(x)
    for a, b in ((om.cons(x), native.cons(x)), (om.grad(x), native.grad(x)), (om.jac_coord(x), native.jac_coord(x)),
                 (om.hess_coord(x, y, 0.6), native.hess_coord(x, y, 0.6))):
        assert np.array_equal(a, b)
    # and the generator accepts it: kernels generated from the foreign encoding, emulated
    em = EmulatedModel(fcore, fblob)
    rel = lambda a, b: 0.0 if b.size == 0 else float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))
    assert abs(em.obj(x) - om.obj(x)) <= 1e-12 * max(1.0, abs(om.obj(x)))
    assert rel(em.cons(x), om.cons(x)) <= 1e-14 and rel(em.grad(x), om.grad(x)) <= 1e-14
    assert rel(em.jac_coord(x, om.nnzj), om.jac_coord(x)) <= 1e-14
    assert rel(em.hess_coord(x, y, 0.6, om.nnzh), om.hess_coord(x, y, 0.6)) <= 1e-14


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["quadrotor_1", "quadrotor_100", "pandemic_20x3", "opf_7", "irregular"])
def test_foreign_encoding_on_gpu(name, grid_mode):
    import torch
    from infiniteexamodels.jl_amd.model import ExaModel
    core = cases.build_core(name)
    _, fblob = foreign_blob(core)
    om = OracleModel(fblob)
    gm = ExaModel.from_blob(fblob, device=0)
    x, y = cases.eval_point_for(name, om, seed=4)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    close = lambda a, b: np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-10 * max(1.0, float(np.abs(b).max()) if b.size else 1.0))
    assert abs(gm.obj(xd) - om.obj(x)) <= 1e-10 * max(1.0, abs(om.obj(x)))
    nan = lambda n: torch.full((n,), float("nan"), device="cuda", dtype=torch.float64)
    close(gm.cons(xd, nan(om.ncon)).cpu().numpy(), om.cons(x))
    close(gm.grad(xd, nan(om.nvar)).cpu().numpy(), om.grad(x))
    close(gm.jac_coord(xd, nan(om.nnzj)).cpu().numpy(), om.jac_coord(x))
    close(gm.hess_coord(xd, yd, nan(om.nnzh), obj_weight=0.6).cpu().numpy(), om.hess_coord(x, y, 0.6))
    r, c = gm.jac_structure()
    ro, co = om.jac_structure()
    assert np.array_equal(np.asarray(r), ro) and np.array_equal(np.asarray(c), co)
    gm.close()


@pytest.mark.parametrize("name", ["pandemic_300x7", "quadrotor_oc3_700", "ode_5x5"])
def test_lattice_recovery_restores_kernel_quality(name, lane_fused):
    """A flat explicit iterator that was an Iterators.product (pandemic's t x xi, the collocation
    node boxes) is recognised by the parser: index columns become affine (never read at run time),
    float columns shrink to one coordinate — the Jacobian kernel reads what the native encoding
    reads (within 25 %), instead of ~10x more."""
    from infiniteexamodels.jl_amd import lib as iemlib

    def jac_read_bytes(blob):
        ks = [l.split() for l in iemlib.emit_launch_plan(blob).splitlines() if l.startswith("kernel")]
        return sum(int(k[k.index("rbytes") + 1]) for k in ks if k[3] == "1")
    core = cases.build_core(name)
    native = jac_read_bytes(core.to_blob())
    foreign = jac_read_bytes(foreign_blob(core)[1])
    assert foreign <= 1.25 * native, (native, foreign)
