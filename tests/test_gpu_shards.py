"""GPU parity of SHARD cores and of the BASELINE.json configs at their named sizes.

A shard of rank > 0 is a different model from the global one (no point constraints / first-stage
rows, a halo row in front of the owned supports, rank-0-only templates missing): every entry
point is evaluated through the C-ABI on the GPU and compared with the (threaded) CPU oracle on
the same shard, indices bit-exact, values within 1e-10 relative.  Sizes: the sharded forms of
BASELINE configs 4 and 5 (OPF 1e4 scenarios, farmer 1e5 scenarios, 8 ranks), config 3
(pandemic 5000 x 100, 8 ranks over xi) and a time-sharded quadrotor; plus the unsharded configs
3, 4, 5 at exactly their named sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def _close(got, ref, what):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, what
    if ref.size == 0:
        return
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    scale = np.maximum(np.abs(ref), 1e-10 * max(1.0, np.abs(ref).max()))
    err = np.abs(got - ref) / scale
    k = int(err.argmax())
    assert err[k] <= RTOL, f"{what}: rel err {err[k]:.3e} at {k} (got {got[k]!r}, ref {ref[k]!r})"


def _build(spec):
    from infiniteexamodels.jl_amd import shard, transcribe, workloads
    kind = spec[0]
    if kind == "quadrotor_shard":
        return shard.quadrotor_shard(*spec[1:])[0], False
    if kind == "opf_shard":
        return shard.opf_shard(*spec[1:])[0], False
    if kind == "farmer_shard":
        return shard.farmer_shard(*spec[1:])[0], True
    if kind == "pandemic_shard":
        return shard.pandemic_shard(*spec[1:])[0], True
    if kind == "quadrotor_oc3_shard":     # cut in the library (the Python transcriber cannot shard collocation models)
        from infiniteexamodels.jl_amd import lib as iemlib

        class _Blob:
            def __init__(self, b):
                self._b = b

            def to_blob(self):
                return self._b
        g = transcribe.exa_core(workloads.quadrotor(spec[1], collocation=3)).to_blob()
        return _Blob(iemlib.shard_blob(g, 1, spec[2], spec[3])[0]), False
    if kind == "pandemic":
        return transcribe.exa_core(workloads.pandemic(*spec[1:])), True
    if kind == "opf":
        return transcribe.exa_core(workloads.opf(*spec[1:])), False
    if kind == "farmer":
        return transcribe.exa_core(workloads.farmer(*spec[1:])), True
    raise KeyError(kind)


SPECS = [
    ("quadrotor_shard", 4000, 0, 4), ("quadrotor_shard", 4000, 1, 4), ("quadrotor_shard", 4000, 3, 4),
    ("quadrotor_shard", 1_000_000, 5, 8),                                # a shard of the headline run
    ("opf_shard", 10_000, 0, 8), ("opf_shard", 10_000, 5, 8),            # config 4, sharded
    ("farmer_shard", 100_000, 0, 8), ("farmer_shard", 100_000, 7, 8),    # config 5, sharded
    ("pandemic_shard", 4990, 100, 0, 8), ("pandemic_shard", 4990, 100, 3, 8),   # config 3 over xi
    ("quadrotor_oc3_shard", 16_000, 0, 4), ("quadrotor_oc3_shard", 16_000, 2, 4),   # the reference ladder's top size (ESCAPE34), collocation
    ("pandemic", 4990, 100),     # config 3 at exactly 5 000 x 100 supports
    ("opf", 10_000),             # config 4, one GPU
    ("farmer", 100_000),         # config 5, one GPU
]


@pytest.mark.parametrize("spec", SPECS, ids=lambda s: "-".join(str(v) for v in s))
def test_shard_and_config_cores_match_oracle(spec, built):
    import torch
    from infiniteexamodels.jl_amd.model import ExaModel
    from pyoracle import OracleModel
    core, positive = _build(spec)
    blob = core.to_blob()
    gm = ExaModel(core if hasattr(core, "templates") else None, device=0, blob=blob)
    om = OracleModel(blob)
    del blob
    om.set_threads(min(om.max_threads(), 32))
    assert (gm.meta.nvar, gm.meta.ncon, gm.meta.nnzj, gm.meta.nnzh) == (om.nvar, om.ncon, om.nnzj, om.nnzh)
    # structure, generated on the device, bit-exact (1-based as Julia reads it)
    r, c = gm.jac_structure_device(1)
    ro, co = om.jac_structure(1)
    assert np.array_equal(r.cpu().numpy(), ro) and np.array_equal(c.cpu().numpy(), co)
    r, c = gm.hess_structure_device(1)
    ro, co = om.hess_structure(1)
    assert np.array_equal(r.cpu().numpy(), ro) and np.array_equal(c.cpu().numpy(), co)
    del r, c, ro, co
    for which in ("x0", "lvar", "uvar", "lcon", "ucon"):
        np.testing.assert_array_equal(getattr(gm.meta, which), getattr(om, which))
    np.testing.assert_array_equal(gm.theta, om.theta)
    x = om.x0 + 0.1 * np.random.default_rng(3).standard_normal(om.nvar)
    if positive:
        x = np.abs(x) + 0.05
    y = np.random.default_rng(4).standard_normal(om.ncon)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    nanv = lambda n: torch.full((n,), float("nan"), device="cuda", dtype=torch.float64)
    f, fo = gm.obj(xd), om.obj(x)
    assert abs(f - fo) <= RTOL * max(1.0, abs(fo)), (f, fo)
    _close(gm.cons(xd, nanv(om.ncon)).cpu().numpy(), om.cons(x), "cons")
    _close(gm.grad(xd, nanv(om.nvar)).cpu().numpy(), om.grad(x), "grad")
    _close(gm.jac_coord(xd, nanv(om.nnzj)).cpu().numpy(), om.jac_coord(x), "jac_coord")
    _close(gm.hess_coord(xd, yd, nanv(om.nnzh), obj_weight=0.6).cpu().numpy(), om.hess_coord(x, y, 0.6), "hess_coord")
    v, vc = np.random.default_rng(5).standard_normal(om.nvar), np.random.default_rng(6).standard_normal(om.ncon)
    vd, vcd = torch.tensor(v, device="cuda"), torch.tensor(vc, device="cuda")
    _close(gm.jprod(xd, vd, nanv(om.ncon)).cpu().numpy(), om.jprod(x, v), "jprod")
    _close(gm.jtprod(xd, vcd, nanv(om.nvar)).cpu().numpy(), om.jtprod(x, vc), "jtprod")
    _close(gm.hprod(xd, yd, vd, nanv(om.nvar), obj_weight=0.6).cpu().numpy(), om.hprod(x, y, v, 0.6), "hprod")
    gm.close()


def test_backend_plug_point_with_a_shard(built):
    """`ExaTranscriptionBackend(solver; backend = MI355XBackend(0, shard = (group, rank, world)))`: the reference's plug
    point (src/infiniteopt_backend.jl:112-131,155-156) on one rank of a sharded run — global transcription, window cut in
    the library, and the parameter-update hook (:511-550 -> set_parameter!) reaching the rank's (global) theta."""
    import torch
    import cases
    from infiniteexamodels.jl_amd import lib as iemlib
    from infiniteexamodels.jl_amd.backend import ExaTranscriptionBackend
    from infiniteexamodels.jl_amd.model import MI355XBackend
    from pyoracle import OracleModel
    m, (P1, P2) = cases.rosenbrock()
    be = ExaTranscriptionBackend(solver=None, backend=MI355XBackend(0, shard=(1, 1, 2)))
    be.build_transformation_backend(m)
    gm = be.model
    info = gm.shard_info()
    assert (info["rank"], info["world"], info["n_global"]) == (1, 2, 3) and gm.meta.nvar < be.core.nvar
    vm, _ = gm.shard_var_map()
    x = np.array([0.4, 0.5, 0.6, 1.9, 2.0, 2.1])[vm]
    xd = torch.tensor(x, device="cuda")
    for p1 in (100.0, 90.0):
        if p1 != 100.0:
            assert be.update_parameter_value(P1, p1)
        so = OracleModel(iemlib.shard_blob(be.core.to_blob(), 1, 1, 2)[0])      # the same cut of the UPDATED core
        np.testing.assert_array_equal(gm.theta, so.theta)
        assert abs(gm.obj(xd) - so.obj(x)) <= RTOL * max(1.0, abs(so.obj(x)))
        _close(gm.grad(xd).cpu().numpy(), so.grad(x), "grad")
        _close(gm.jac_coord(xd).cpu().numpy(), so.jac_coord(x), "jac")
    # the sharded plug point is build-and-evaluate only: the single-process solver slot is refused loudly
    # (a global x0 must never reach a local model)
    be.solver = lambda *a, **k: None
    with pytest.raises(NotImplementedError, match="rank 1 of 2"):
        be.optimize()
    with pytest.raises(NotImplementedError):
        be.warmstart_backend_start_values()
    be.empty()


@pytest.mark.parametrize("rank,world", [(0, 3), (1, 3), (2, 3)])
def test_sharded_handle_with_an_explicit_item_list(rank, world, built):
    """iem_create_sharded on a model with a domain-restricted constraint (explicit item list, index columns):
    the rank's filtered list evaluates on the GPU like the oracle on the device-free cut of the same shard."""
    import torch
    from infiniteexamodels.jl_amd import lib as iemlib, transcribe
    from infiniteexamodels.jl_amd.model import ExaModel
    from pyoracle import OracleModel
    from test_shard_cabi import _restricted
    gblob = transcribe.exa_core(_restricted(n=3000)).to_blob()
    gm = ExaModel.sharded(gblob, 1, rank, world, device=0)
    lblob, info, vmap, vflag, tpl = iemlib.shard_blob(gblob, 1, rank, world)
    om = OracleModel(lblob)
    assert any(t["items_offset"] >= 0 for t in gm.shard_templates())
    assert [t["ordinals"].tolist() for t in gm.shard_templates()] == [t["ordinals"].tolist() for t in tpl]
    x = np.abs(om.x0 + 0.1 * np.random.default_rng(3).standard_normal(om.nvar)) + 0.05
    y = np.random.default_rng(4).standard_normal(om.ncon)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    r, c = gm.jac_structure_device(1)
    ro, co = om.jac_structure(1)
    assert np.array_equal(r.cpu().numpy(), ro) and np.array_equal(c.cpu().numpy(), co)
    _close(gm.cons(xd).cpu().numpy(), om.cons(x), "cons")
    _close(gm.jac_coord(xd).cpu().numpy(), om.jac_coord(x), "jac")
    _close(gm.hess_coord(xd, yd, obj_weight=0.6).cpu().numpy(), om.hess_coord(x, y, 0.6), "hess")
    _close(gm.grad(xd).cpu().numpy(), om.grad(x), "grad")
    assert abs(gm.obj(xd) - om.obj(x)) <= RTOL * max(1.0, abs(om.obj(x)))
    gm.close()
