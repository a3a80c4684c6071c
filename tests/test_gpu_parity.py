"""GPU parity tests: HIP path (through the C-ABI) vs the CPU oracle on the same seeded
inputs.  Bar (BASELINE.json north_star): sparsity indices bit-exact, Float64 values
within 1e-10 relative."""
import numpy as np
import pytest

import cases
from helpers import coo_to_dense  # noqa: F401

pytestmark = pytest.mark.gpu

RTOL = 1e-10   # north_star: "within 1e-10 rel on Float64 values"


def _close(got, ref, what):
    got = np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, what
    if ref.size == 0:
        return
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    scale = np.maximum(np.abs(ref), 1e-10 * max(1.0, np.abs(ref).max()))
    err = np.abs(got - ref) / scale
    k = int(err.argmax())
    assert err[k] <= RTOL, f"{what}: rel err {err[k]:.3e} at {k} (got {got[k]!r}, ref {ref[k]!r})"


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _models(name, torch):
    from infiniteexamodels.jl_amd.model import ExaModel
    from pyoracle import OracleModel
    core = cases.build_core(name)
    blob = core.to_blob()
    return core, OracleModel(blob), ExaModel(core, device=0, blob=blob)


@pytest.mark.parametrize("name", list(cases.small_cases()))
def test_all_entry_points_match_oracle(name, torch_cuda, grid_mode):
    torch = torch_cuda
    core, om, gm = _models(name, torch)
    assert (gm.meta.nvar, gm.meta.ncon, gm.meta.nnzj, gm.meta.nnzh) == (om.nvar, om.ncon, om.nnzj, om.nnzh)
    for i in range(om.n_templates):
        assert gm.template_info(i) == om.template_info(i)
    # structure: bit-exact
    for base in (0, 1):
        r, c = gm.jac_structure(base)
        ro, co = om.jac_structure(base)
        assert np.array_equal(r, ro) and np.array_equal(c, co)
        r, c = gm.hess_structure(base)
        ro, co = om.hess_structure(base)
        assert np.array_equal(r, ro) and np.array_equal(c, co)
    for base in (0, 1):   # device-generated structure == host structure == oracle
        r, c = gm.jac_structure_device(base)
        ro, co = om.jac_structure(base)
        assert np.array_equal(r.cpu().numpy(), ro) and np.array_equal(c.cpu().numpy(), co)
        r, c = gm.hess_structure_device(base)
        ro, co = om.hess_structure(base)
        assert np.array_equal(r.cpu().numpy(), ro) and np.array_equal(c.cpu().numpy(), co)
    np.testing.assert_array_equal(gm.meta.x0, om.x0)
    np.testing.assert_array_equal(gm.meta.lvar, om.lvar)
    np.testing.assert_array_equal(gm.meta.uvar, om.uvar)
    np.testing.assert_array_equal(gm.meta.lcon, om.lcon)
    np.testing.assert_array_equal(gm.meta.ucon, om.ucon)
    np.testing.assert_array_equal(gm.theta, om.theta)
    for seed in (0, 7):
        x, y = cases.eval_point_for(name, om, seed)
        xd = torch.tensor(x, device="cuda")
        yd = torch.tensor(y, device="cuda")
        f = gm.obj(xd)
        fo = om.obj(x)
        assert abs(f - fo) <= RTOL * max(1.0, abs(fo)), (f, fo)
        # outputs must be fully overwritten: poison the buffers first
        cv = torch.full((om.ncon,), float("nan"), device="cuda", dtype=torch.float64)
        gv = torch.full((om.nvar,), float("nan"), device="cuda", dtype=torch.float64)
        _close(gm.cons(xd, cv).cpu().numpy(), om.cons(x), "cons")
        _close(gm.grad(xd, gv).cpu().numpy(), om.grad(x), "grad")
        jv = torch.full((om.nnzj,), float("nan"), device="cuda", dtype=torch.float64)
        hv = torch.full((om.nnzh,), float("nan"), device="cuda", dtype=torch.float64)
        _close(gm.jac_coord(xd, jv).cpu().numpy(), om.jac_coord(x), "jac_coord")
        for w in (1.0, 0.3):
            _close(gm.hess_coord(xd, yd, hv, obj_weight=w).cpu().numpy(), om.hess_coord(x, y, w), "hess_coord")
        # the fused launch (iem_jac_hess_coord) into poisoned buffers: the BYTES of the two calls (w = 0.3: the last above)
        jv2 = torch.full((om.nnzj,), float("nan"), device="cuda", dtype=torch.float64)
        hv2 = torch.full((om.nnzh,), float("nan"), device="cuda", dtype=torch.float64)
        gm.jac_hess_coord(xd, yd, jv2, hv2, obj_weight=0.3)
        assert torch.equal(jv2, jv) and torch.equal(hv2, hv), "fused jac + hess launch differs from the two calls"
        # one launch per solver phase into poisoned buffers: the BYTES of the separate calls (iem_eval_trial: obj + cons!;
        # iem_eval_accepted: grad! + jac_coord! + hess_coord!, w = 0.3 as the last hess_coord! above)
        cv2 = torch.full((om.ncon,), float("nan"), device="cuda", dtype=torch.float64)
        f2, _ = gm.eval_trial(xd, cv2)
        assert f2 == f and torch.equal(cv2, cv), "iem_eval_trial differs from obj + cons!"
        _, cv3 = gm.eval_trial(xd, torch.full((om.ncon,), float("nan"), device="cuda", dtype=torch.float64), defer_obj=True)
        assert gm.obj_end() == f and torch.equal(cv3, cv)
        gv2, jv3, hv3 = (torch.full((n_,), float("nan"), device="cuda", dtype=torch.float64) for n_ in (om.nvar, om.nnzj, om.nnzh))
        gm.eval_accepted(xd, yd, gv2, jv3, hv3, obj_weight=0.3)
        assert torch.equal(gv2, gv) and torch.equal(jv3, jv) and torch.equal(hv3, hv), "iem_eval_accepted differs from the three calls"
        cv4, gv4, jv4, hv4 = (torch.full((n_,), float("nan"), device="cuda", dtype=torch.float64) for n_ in (om.ncon, om.nvar, om.nnzj, om.nnzh))
        f4 = gm.eval_all(xd, yd, cv4, gv4, jv4, hv4, obj_weight=0.3)[0]
        assert f4 == f and torch.equal(cv4, cv) and torch.equal(gv4, gv) and torch.equal(jv4, jv) and torch.equal(hv4, hv), "iem_eval_all differs from the five calls"
        # obj in two halves (iem_obj_begin / iem_obj_end) around other launches: the same bits as obj
        gm.obj_begin(xd)
        gm.cons(xd, cv)
        assert gm.obj_end() == f
        # matrix-free products into poisoned buffers
        rng = np.random.default_rng(seed + 40)
        v, vc = rng.standard_normal(om.nvar), rng.standard_normal(om.ncon)
        vd, vcd = torch.tensor(v, device="cuda"), torch.tensor(vc, device="cuda")
        nanv = lambda n: torch.full((n,), float("nan"), device="cuda", dtype=torch.float64)
        _close(gm.jprod(xd, vd, nanv(om.ncon)).cpu().numpy(), om.jprod(x, v), "jprod")
        _close(gm.jtprod(xd, vcd, nanv(om.nvar)).cpu().numpy(), om.jtprod(x, vc), "jtprod")
        _close(gm.hprod(xd, yd, vd, nanv(om.nvar), obj_weight=0.3).cpu().numpy(), om.hprod(x, y, v, 0.3), "hprod")
    gm.close()


@pytest.mark.parametrize("name", ["quadrotor_1000", "pandemic_300x7", "irregular", "test_problem_1"])
def test_store_modes_agree(name, torch_cuda, grid_mode):
    """Direct strided stores (0), wave-level LDS-transposed stores (1) and the
    block-cooperative 128-byte-aligned stores (2) write identical bytes."""
    torch = torch_cuda
    from infiniteexamodels.jl_amd import lib as iemlib
    from infiniteexamodels.jl_amd.model import ExaModel
    from pyoracle import OracleModel
    core = cases.build_core(name)
    blob = core.to_blob()
    om = OracleModel(blob)
    x, y = cases.eval_point_for(name, om, 3)
    out = {}
    try:
        for mode in (0, 1, 2):
            iemlib.set_option("store_mode", mode)
            gm = ExaModel(core, device=0, blob=blob)
            xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
            jv = torch.full((om.nnzj,), float("nan"), device="cuda", dtype=torch.float64)
            hv = torch.full((om.nnzh,), float("nan"), device="cuda", dtype=torch.float64)
            out[mode] = (gm.jac_coord(xd, jv).cpu().numpy(), gm.hess_coord(xd, yd, hv).cpu().numpy())
            gm.close()
    finally:
        iemlib.set_option("store_mode", iemlib.DEFAULT_STORE_MODE)
    for mode in (1, 2):
        assert np.array_equal(out[0][0], out[mode][0]), f"jac differs in store_mode {mode}"
        assert np.array_equal(out[0][1], out[mode][1]), f"hess differs in store_mode {mode}"
    _close(out[2][0], om.jac_coord(x), "jac")
    _close(out[2][1], om.hess_coord(x, y, 1.0), "hess")


def test_set_parameter_updates_theta(torch_cuda):
    """ExaModels.set_parameter! (infiniteopt_backend.jl:522,546) reaches the device copy of θ."""
    torch = torch_cuda
    from infiniteexamodels.jl_amd import transcribe
    from infiniteexamodels.jl_amd.model import ExaModel
    from pyoracle import OracleModel
    m, (P1, P2) = cases.rosenbrock()
    data = transcribe.ExaMappingData()
    core = transcribe.exa_core(m, data)
    gm = ExaModel(core, device=0)
    om = OracleModel(core.to_blob())
    x = np.array([0.4, 0.5, 0.6, 1.9, 2.0, 2.1])
    xd = torch.tensor(x, device="cuda")
    assert abs(gm.obj(xd) - om.obj(x)) < 1e-10 * abs(om.obj(x))
    core.set_parameter(data.param_mappings[P1], [90.0])
    core.set_parameter(data.param_mappings[P2], [1.3])
    om.set_parameter(data.param_mappings[P1].offset, [90.0])
    om.set_parameter(data.param_mappings[P2].offset, [1.3])
    np.testing.assert_array_equal(gm.theta, [90.0, 1.3])
    assert abs(gm.obj(xd) - om.obj(x)) < 1e-10 * abs(om.obj(x))
    _close(gm.grad(xd).cpu().numpy(), om.grad(x), "grad after set_parameter")
    gm.close()


def test_quadrotor_1e5_properties(torch_cuda):
    """Config 2 of BASELINE.json at full size: compare against the oracle on the whole
    problem (the C oracle finishes 1e5 supports in about a second) and check
    size-independent properties: linearity of hess_coord in (obj_weight, y) and
    idempotence of repeated evaluation."""
    torch = torch_cuda
    from infiniteexamodels.jl_amd import transcribe, workloads
    from infiniteexamodels.jl_amd.model import ExaModel
    from pyoracle import OracleModel
    S = 100_000
    core = transcribe.exa_core(workloads.quadrotor(S))
    blob = core.to_blob()
    gm = ExaModel(core, device=0, blob=blob)
    om = OracleModel(blob)
    om.set_threads(om.max_threads())
    assert gm.meta.nvar == 22 * S and gm.meta.ncon == 18 * S and gm.meta.nnzj == 62 * S - 18
    rng = np.random.default_rng(0)
    x = gm.meta.x0 + 0.1 * rng.standard_normal(gm.meta.nvar)
    x[7 * S:8 * S] = np.clip(x[7 * S:8 * S], -1.2, 1.2)
    y = np.random.default_rng(1).standard_normal(gm.meta.ncon)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    j1 = gm.jac_coord(xd).cpu().numpy()
    _close(j1, om.jac_coord(x), "jac_coord 1e5")
    h1 = gm.hess_coord(xd, yd, obj_weight=1.0).cpu().numpy()
    _close(h1, om.hess_coord(x, y, 1.0), "hess_coord 1e5")
    # idempotence
    assert np.array_equal(j1, gm.jac_coord(xd).cpu().numpy())
    assert np.array_equal(h1, gm.hess_coord(xd, yd, obj_weight=1.0).cpu().numpy())
    # linearity in (obj_weight, y): H(2w, 2y) == 2 H(w, y) exactly (power-of-two scaling)
    h2 = gm.hess_coord(xd, 2 * yd, obj_weight=2.0).cpu().numpy()
    assert np.array_equal(h2, 2 * h1)
    # the one-launch pair at this size: the bytes of the two calls, into poisoned buffers
    jp = torch.full((gm.meta.nnzj,), float("nan"), device="cuda", dtype=torch.float64)
    hp = torch.full((gm.meta.nnzh,), float("nan"), device="cuda", dtype=torch.float64)
    gm.jac_hess_coord(xd, yd, jp, hp, obj_weight=1.0)
    assert np.array_equal(jp.cpu().numpy(), j1) and np.array_equal(hp.cpu().numpy(), h1)
    # structure equals the oracle's, position by position
    r, c = gm.jac_structure()
    ro, co = om.jac_structure()
    assert np.array_equal(r, ro) and np.array_equal(c, co)
    r, c = gm.hess_structure()
    ro, co = om.hess_structure()
    assert np.array_equal(r, ro) and np.array_equal(c, co)
    gm.close()


def test_quadrotor_1e6_headline_size(torch_cuda):
    """The headline size of BASELINE.json (10^6 supports): the GPU result is compared with
    the oracle over the WHOLE problem (the C oracle, all host threads, needs a few seconds)
    plus a checksum-of-blocks property: per-template sums of the COO block equal the oracle's."""
    torch = torch_cuda
    from infiniteexamodels.jl_amd import transcribe, workloads
    from infiniteexamodels.jl_amd.model import ExaModel
    from pyoracle import OracleModel
    S = 1_000_000
    core = transcribe.exa_core(workloads.quadrotor(S))
    blob = core.to_blob()
    gm = ExaModel(core, device=0, blob=blob)
    om = OracleModel(blob)
    del blob
    om.set_threads(min(om.max_threads(), 32))
    x = gm.meta.x0 + 0.1 * np.random.default_rng(0).standard_normal(gm.meta.nvar)
    x[7 * S:8 * S] = np.clip(x[7 * S:8 * S], -1.2, 1.2)
    y = np.random.default_rng(1).standard_normal(gm.meta.ncon)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    j = gm.jac_coord(xd).cpu().numpy()
    jo = om.jac_coord(x)
    _close(j, jo, "jac_coord 1e6")
    h = gm.hess_coord(xd, yd, obj_weight=1.0).cpu().numpy()
    ho = om.hess_coord(x, y, 1.0)
    _close(h, ho, "hess_coord 1e6")
    for i in range(om.n_templates):
        t = om.template_info(i)
        if t["o2step"]:
            a, b = t["o2"], t["o2"] + t["n_items"] * t["o2step"]
            assert abs(h[a:b].sum() - ho[a:b].sum()) <= 1e-9 * max(1.0, np.abs(ho[a:b]).sum())
    r, c = gm.jac_structure_device()
    ro, co = om.jac_structure()
    assert np.array_equal(r.cpu().numpy(), ro) and np.array_equal(c.cpu().numpy(), co)
    # the bench's default step at this size (iem_jac_hess_coord) writes the same bytes; so does the LARGE-GRID kernel shape
    # (what a model of >= 2e6 supports gets: 48-slot staging batch + XCD-aware tile walk), forced here by its thresholds
    jp = torch.full((gm.meta.nnzj,), float("nan"), device="cuda", dtype=torch.float64)
    hp = torch.full((gm.meta.nnzh,), float("nan"), device="cuda", dtype=torch.float64)
    gm.jac_hess_coord(xd, yd, jp, hp, obj_weight=1.0)
    assert np.array_equal(jp.cpu().numpy(), j) and np.array_equal(hp.cpu().numpy(), h)
    del jp, hp
    big = ExaModel(core, device=0, options=dict(big_batch_jac=1000, big_batch_hess=1000))
    assert {k["kind"]: k["lds_bytes"] for k in big.kernels() if k["kind"] in ("jac", "hess")} == {"jac": 98304, "hess": 98304}
    assert np.array_equal(big.jac_coord(xd).cpu().numpy(), j) and np.array_equal(big.hess_coord(xd, yd, obj_weight=1.0).cpu().numpy(), h)
    big.close(); gm.close()


def test_eval_loop_is_graph_capturable(torch_cuda, grid_mode):
    """The five-call evaluation loop captured into a HIP graph (torch.cuda.graph) and replayed
    on new inputs gives the same results as eager calls: no allocation / synchronisation
    hides in the launch path."""
    torch = torch_cuda
    from infiniteexamodels.jl_amd.model import ExaModel
    from pyoracle import OracleModel
    core = cases.build_core("pandemic_300x7")
    blob = core.to_blob()
    gm = ExaModel(core, device=0, blob=blob)
    om = OracleModel(blob)
    x0, y0 = cases.eval_point_for("pandemic_300x7", om, 0)
    x1, y1 = cases.eval_point_for("pandemic_300x7", om, 5)
    xd, yd = torch.tensor(x0, device="cuda"), torch.tensor(y0, device="cuda")
    f = torch.zeros(1, dtype=torch.float64, device="cuda")
    g = torch.empty(om.nvar, dtype=torch.float64, device="cuda")
    c = torch.empty(om.ncon, dtype=torch.float64, device="cuda")
    jv = torch.empty(om.nnzj, dtype=torch.float64, device="cuda")
    hv = torch.empty(om.nnzh, dtype=torch.float64, device="cuda")

    def loop():
        gm.obj_device(xd, f); gm.grad(xd, g); gm.cons(xd, c); gm.jac_coord(xd, jv); gm.hess_coord(xd, yd, hv, obj_weight=0.4)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        loop()                      # warm-up outside capture
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loop()
    xd.copy_(torch.tensor(x1)); yd.copy_(torch.tensor(y1))
    for out in (f, g, c, jv, hv):
        out.fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    assert abs(f.item() - om.obj(x1)) <= 1e-10 * max(1.0, abs(om.obj(x1)))
    _close(g.cpu().numpy(), om.grad(x1), "grad (graph)")
    _close(c.cpu().numpy(), om.cons(x1), "cons (graph)")
    _close(jv.cpu().numpy(), om.jac_coord(x1), "jac (graph)")
    _close(hv.cpu().numpy(), om.hess_coord(x1, y1, 0.4), "hess (graph)")
    gm.close()


@pytest.mark.parametrize("name", ["quadrotor_1000", "pandemic_300x7", "opf_600", "irregular"])
def test_merged_hessian_layout_gpu(name, torch_cuda):
    """hess_layout="merged": structure (host and device) and values are mutually consistent and
    sum to the oracle's Hessian."""
    torch = torch_cuda
    import scipy.sparse as sp
    from infiniteexamodels.jl_amd.model import ExaModel
    from pyoracle import OracleModel
    core = cases.build_core(name)
    blob = core.to_blob()
    om = OracleModel(blob)
    gm = ExaModel(core, device=0, blob=blob, hess_layout="merged")
    assert gm.meta.nnzh < om.nnzh
    x, y = cases.eval_point_for(name, om)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    hv = torch.full((gm.meta.nnzh,), float("nan"), device="cuda", dtype=torch.float64)
    h = gm.hess_coord(xd, yd, hv, obj_weight=0.7).cpu().numpy()
    r, c = gm.hess_structure()
    rd, cd = gm.hess_structure_device()
    assert np.array_equal(rd.cpu().numpy(), r) and np.array_equal(cd.cpu().numpy(), c) and (r >= c).all()
    ro, co = om.hess_structure()
    A = sp.coo_matrix((h, (r, c)), shape=(om.nvar, om.nvar)).tocsr()
    B = sp.coo_matrix((om.hess_coord(x, y, 0.7), (ro, co)), shape=(om.nvar, om.nvar)).tocsr()
    d = abs(A - B)
    assert (d.max() if d.nnz else 0.0) <= 1e-10 * max(1.0, abs(B).max())
    # jac / cons / hprod are unaffected by the Hessian layout
    _close(gm.jac_coord(xd).cpu().numpy(), om.jac_coord(x), "jac")
    v = np.random.default_rng(2).standard_normal(om.nvar)
    _close(gm.hprod(xd, yd, torch.tensor(v, device="cuda"), obj_weight=0.7).cpu().numpy(), om.hprod(x, y, v, 0.7), "hprod")
    gm.close()


@pytest.mark.parametrize("nt,nxi,flat2d", [(54, 66000, 0), (2, 70000, 0), (54, 66000, 1), (490, 2300, 1), (1190, 701, 1)])
def test_folded_or_flat_grids_gpu(nt, nxi, flat2d, torch_cuda):
    """flat2d = 0 — 64 x 66000 supports: the second grid dimension is folded over blockIdx.z; 12 x 70000: the
    short first dimension switches the kernel to flat lane indexing.  flat2d = 1 (default): every 2-D grid is
    walked by one linear lane index and sub-box templates are stored by item ordinal (iem_flush_ord)."""
    torch = torch_cuda
    from infiniteexamodels.jl_amd import transcribe, workloads
    from infiniteexamodels.jl_amd.model import ExaModel
    from pyoracle import OracleModel
    core = transcribe.exa_core(workloads.pandemic(nt, nxi))
    blob = core.to_blob()
    om = OracleModel(blob)
    om.set_threads(min(16, om.max_threads()))
    gm = ExaModel(core, device=0, blob=blob, options={"flat2d": flat2d})
    x = np.abs(om.x0 + 0.1 * np.random.default_rng(0).standard_normal(om.nvar)) + 0.05
    y = np.random.default_rng(1).standard_normal(om.ncon)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    nanv = lambda n: torch.full((n,), float("nan"), device="cuda", dtype=torch.float64)
    _close(gm.cons(xd, nanv(om.ncon)).cpu().numpy(), om.cons(x), "cons")
    _close(gm.jac_coord(xd, nanv(om.nnzj)).cpu().numpy(), om.jac_coord(x), "jac")
    _close(gm.hess_coord(xd, yd, nanv(om.nnzh)).cpu().numpy(), om.hess_coord(x, y, 1.0), "hess")
    assert abs(gm.obj(xd) - om.obj(x)) <= 1e-10 * max(1.0, abs(om.obj(x)))
    gm.close()


def test_corrupt_cached_code_object_is_rebuilt(torch_cuda, tmp_path, monkeypatch):
    """A cached code object that does not load (truncated / foreign file) is recompiled by hiprtc
    instead of failing the model."""
    import os
    torch = torch_cuda
    from infiniteexamodels.jl_amd import lib as iemlib
    from infiniteexamodels.jl_amd.model import ExaModel
    from pyoracle import OracleModel
    cache = tmp_path / "cache"
    cache.mkdir()
    monkeypatch.setenv("IEM_KERNEL_CACHE", str(cache))
    core = cases.build_core("quadrotor_5")
    blob = core.to_blob()
    _, key = iemlib.emit_source(blob)
    gm = ExaModel(core, device=0, blob=blob)          # JIT, populates the private cache
    gm.close()
    files = [f for f in os.listdir(cache) if f.endswith(".hsaco")]
    assert len(files) == 1
    (cache / files[0]).write_bytes(b"not a code object")
    gm = ExaModel(core, device=0, blob=blob)
    om = OracleModel(blob)
    x, _ = cases.eval_point_for("quadrotor_5", om)
    np.testing.assert_allclose(gm.jac_coord(torch.tensor(x, device="cuda")).cpu().numpy(), om.jac_coord(x), rtol=1e-12, atol=1e-12)
    assert (cache / files[0]).stat().st_size > 1000   # rewritten
    gm.close()


def test_two_handles_with_different_options_coexist(torch_cuda):
    """Per-handle options (iem_create_opts): a merged-Hessian model, a default one and one with the
    wave-level store path live side by side in one process — no process-global knob decides what a
    handle generates — and interleaved calls give each handle's own layout."""
    torch = torch_cuda
    from infiniteexamodels.jl_amd import lib as iemlib
    from infiniteexamodels.jl_amd.model import ExaModel
    from pyoracle import OracleModel
    core = cases.build_core("quadrotor_1000")
    blob = core.to_blob()
    om = OracleModel(blob)
    x, y = cases.eval_point_for("quadrotor_1000", om, 2)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    merged = ExaModel(core, device=0, blob=blob, hess_layout="merged")
    plain = ExaModel(core, device=0, blob=blob)
    waves = ExaModel(core, device=0, blob=blob, options={"store_mode": 1, "split_small": 0, "obj_wgs": 3})
    assert merged.meta.nnzh < plain.meta.nnzh == waves.meta.nnzh == om.nnzh
    # the process defaults are untouched by any of this
    src_before, key_before = iemlib.emit_source(blob)
    for _ in range(2):
        hm = merged.hess_coord(xd, yd, obj_weight=0.7).cpu().numpy()
        hp = plain.hess_coord(xd, yd, obj_weight=0.7).cpu().numpy()
        hw = waves.hess_coord(xd, yd, obj_weight=0.7).cpu().numpy()
        assert hm.shape[0] == merged.meta.nnzh
        _close(hp, om.hess_coord(x, y, 0.7), "hess (default handle)")
        assert np.array_equal(hp, hw)
        assert abs(waves.obj(xd) - om.obj(x)) <= 1e-10 * max(1.0, abs(om.obj(x)))
        assert abs(plain.obj(xd) - om.obj(x)) <= 1e-10 * max(1.0, abs(om.obj(x)))
    r, c = merged.hess_structure()
    import scipy.sparse as sp
    ro, co = om.hess_structure()
    A = sp.coo_matrix((hm, (r, c)), shape=(om.nvar, om.nvar)).tocsr()
    B = sp.coo_matrix((om.hess_coord(x, y, 0.7), (ro, co)), shape=(om.nvar, om.nvar)).tocsr()
    d = abs(A - B)
    assert (d.max() if d.nnz else 0.0) <= 1e-10 * max(1.0, abs(B).max())
    assert iemlib.emit_source(blob)[1] == key_before
    for m in (merged, plain, waves):
        m.close()


def test_offline_code_objects_are_used(torch_cuda):
    """A model whose kernels build() compiled offline (hipcc --genco into infiniteexamodels.jl_amd/kernels) must
    LOAD that object at iem_create, not recompile it with hiprtc (round 1 looked for the cache one directory too
    high and compiled every model at run time)."""
    import os
    from infiniteexamodels.jl_amd import lib as iemlib
    from infiniteexamodels.jl_amd.model import ExaModel
    blob = cases.build_core("quadrotor_100").to_blob()
    _, key = iemlib.emit_source(blob)
    path = os.path.join(iemlib.KERNEL_DIR, f"iem_{key:016x}.hsaco")
    if not os.path.exists(path):
        pytest.skip("no offline-built object for this model in the tree (build() not run)")
    gm = ExaModel.from_blob(blob)
    assert not any(k["jit"] for k in gm.kernels()), "the offline-built code object was not used"
    gm.close()


def test_store_batch_tuner_is_invisible_in_the_results(torch_cuda):
    """Large jac/hess grids keep a second code object (lds_slots = 48) and pick per output buffer from the first twenty
    calls: every call, whichever variant ran, writes the same bytes as a handle with the tuner off, into either of
    two buffers, and the later calls (decided) too."""
    torch = torch_cuda
    from infiniteexamodels.jl_amd import transcribe, workloads
    from infiniteexamodels.jl_amd.model import ExaModel
    from pyoracle import OracleModel
    S = 260_000                                   # 525 workgroups: above autotune_min_blocks
    core = transcribe.exa_core(workloads.quadrotor(S))
    blob = core.to_blob()
    tuned = ExaModel(core, device=0, blob=blob, options={"autotune": 1})     # opt-in
    plain = ExaModel(core, device=0, blob=blob)
    om = OracleModel(blob)
    om.set_threads(min(om.max_threads(), 16))
    x = tuned.meta.x0 + 0.1 * np.random.default_rng(0).standard_normal(tuned.meta.nvar)
    x[7 * S:8 * S] = np.clip(x[7 * S:8 * S], -1.2, 1.2)
    y = np.random.default_rng(1).standard_normal(tuned.meta.ncon)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    jref = plain.jac_coord(xd).cpu().numpy()
    href = plain.hess_coord(xd, yd, obj_weight=0.7).cpu().numpy()
    _close(jref, om.jac_coord(x), "jac")
    _close(href, om.hess_coord(x, y, 0.7), "hess")
    bufs = [(torch.empty(tuned.meta.nnzj, dtype=torch.float64, device="cuda"), torch.empty(tuned.meta.nnzh, dtype=torch.float64, device="cuda")) for _ in range(2)]
    for it in range(24):
        for jb, hb in bufs:                       # each buffer has its own tuner slot
            jb.fill_(float("nan")); hb.fill_(float("nan"))
            assert np.array_equal(tuned.jac_coord(xd, jb).cpu().numpy(), jref), f"jac call {it}"
            assert np.array_equal(tuned.hess_coord(xd, yd, hb, obj_weight=0.7).cpu().numpy(), href), f"hess call {it}"
    for jb, hb in bufs:                           # 24 synchronised calls each: both buffers are decided, separately
        assert tuned.tuner_choice("jac", jb) in (0, 1) and tuned.tuner_choice("hess", hb) in (0, 1)
    assert plain.tuner_choice("jac", bufs[0][0]) == -1
    jb, hb = bufs[0]
    for it in range(12):                          # decided: the chosen variant alone
        jb.fill_(float("nan"))
        assert np.array_equal(tuned.jac_coord(xd, jb).cpu().numpy(), jref), f"jac call {it} (same buffer)"
        hb.fill_(float("nan"))
        assert np.array_equal(tuned.hess_coord(xd, yd, hb, obj_weight=0.7).cpu().numpy(), href)
    # explicit set-up form: iem_tune decides for fresh buffers at once, and what it leaves in them is an evaluation
    jb2 = torch.full((tuned.meta.nnzj,), float("nan"), dtype=torch.float64, device="cuda")
    hb2 = torch.full((tuned.meta.nnzh,), float("nan"), dtype=torch.float64, device="cuda")
    ch = tuned.tune(xd, yd, jb2, hb2, obj_weight=0.7)
    assert ch["jac"] in (0, 1) and ch["hess"] in (0, 1)
    assert np.array_equal(jb2.cpu().numpy(), jref) and np.array_equal(hb2.cpu().numpy(), href)
    assert plain.tune(xd, yd, jb2, hb2) == {"jac": -1, "hess": -1}
    tuned.close(); plain.close()


def test_misuse_is_an_error_code_not_a_crash(torch_cuda):
    """C-ABI misuse on a live handle comes back as a negative code + message (the wrapper raises IemError): the
    multi-GPU calls on an unsharded handle, null / missing arguments, an unknown per-handle option — and the handle
    still evaluates afterwards."""
    torch = torch_cuda
    from infiniteexamodels.jl_amd import lib as iemlib
    from infiniteexamodels.jl_amd.model import ExaModel
    from pyoracle import OracleModel
    core = cases.build_core("quadrotor_100")
    blob = core.to_blob()
    gm = ExaModel(core, device=0, blob=blob)
    om = OracleModel(blob)
    x, y = cases.eval_point_for("quadrotor_100", om)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    g = torch.empty(gm.meta.nvar, dtype=torch.float64, device="cuda")
    L = iemlib.lib()
    for call in (lambda: gm.halo_exchange(xd), lambda: gm.halo_fold(g), lambda: gm.allreduce_obj_grad(None, g), lambda: gm.comm_export()):
        with pytest.raises(iemlib.IemError):
            call()
    assert L.iem_jac_coord(gm._h, None, None) < 0
    assert L.iem_tune(gm._h, xd.data_ptr(), None, 1.0, None, g.data_ptr()) < 0          # a Hessian buffer without y
    assert L.iem_csr_values32(gm._h, 5, None, None, None, None) < 0
    with pytest.raises(KeyError):          # the wrapper knows the option names; the C-ABI itself answers IEM_E_ARG (tests/test_abi.py)
        ExaModel(core, device=0, blob=blob, options={"no_such_option": 1})
    _close(gm.jac_coord(xd).cpu().numpy(), om.jac_coord(x), "jac after misuse")
    gm.close()


def test_obj_begin_end_protocol(torch_cuda):
    """One begin outstanding per handle; end without begin is an error; a model without objective returns 0."""
    torch = torch_cuda
    from infiniteexamodels.jl_amd import lib as iemlib
    core, om, gm = _models("quadrotor_100", torch)
    xd = torch.tensor(om.x0 + 0.1, device="cuda")
    with pytest.raises(iemlib.IemError, match="without iem_obj_begin"):
        gm.obj_end()
    gm.obj_begin(xd)
    with pytest.raises(iemlib.IemError, match="has not been collected"):
        gm.obj_begin(xd)
    assert gm.obj_end() == gm.obj(xd)
    gm.close()


@pytest.mark.parametrize("opts", [dict(big_batch_jac=1, big_batch_hess=0), dict(big_batch_jac=1, big_batch_hess=1), dict(big_batch_jac=1, big_batch_hess=1, big_xcd=0),
                                  dict(big_batch_jac=1, big_batch_hess=1, big_tile=0), dict(xcd_remap=1), dict(pair_kernel=0)])
def test_large_grid_shapes_and_pair_fallback_on_gpu(opts, torch_cuda):
    """The large-grid kernel shape (staging batch for jac only / both, with and without the XCD-aware walk), the XCD remap
    everywhere, and a handle WITHOUT the fused kernel (iem_jac_hess_coord then makes
    the two calls) write the bytes of the default handle."""
    torch = torch_cuda
    from infiniteexamodels.jl_amd.model import ExaModel
    core, om, gm = _models("quadrotor_1000", torch)
    x, y = cases.eval_point_for("quadrotor_1000", om)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    j0, h0 = gm.jac_coord(xd), gm.hess_coord(xd, yd, obj_weight=0.7)
    g2 = ExaModel(core, device=0, options=dict(split_small=0, **opts))
    kinds = [k["kind"] for k in g2.kernels()]
    # the fused launch exists unless switched off — or unless jac_coord! and hess_coord! run different workgroup sizes
    # (large-grid shape for one kind only): a launch has ONE size, iem_jac_hess_coord then makes the two calls
    assert ("pair" in kinds) == (opts.get("pair_kernel", 1) == 1 and opts.get("big_batch_hess", 1) != 0)
    assert torch.equal(g2.jac_coord(xd), j0) and torch.equal(g2.hess_coord(xd, yd, obj_weight=0.7), h0)
    nan = lambda n: torch.full((n,), float("nan"), device="cuda", dtype=torch.float64)
    j2, h2 = g2.jac_hess_coord(xd, yd, nan(om.nnzj), nan(om.nnzh), obj_weight=0.7)
    assert torch.equal(j2, j0) and torch.equal(h2, h0)
    _close(j2.cpu().numpy(), om.jac_coord(x), "jac"); _close(h2.cpu().numpy(), om.hess_coord(x, y, 0.7), "hess")
    g2.close(); gm.close()


@pytest.mark.parametrize("name", ["quadrotor_1000", "pandemic_300x7", "farmer_1000", "opf_600"])
def test_raw_loop_forms_agree(name, torch_cuda):
    """ExaModel.raw_loop — the five evaluations of one solver point through prebound C calls: plain order, objective
    deferred (iem_obj_begin first / iem_obj_end last), and deferred + one-launch pair — same value, same bytes."""
    torch = torch_cuda
    core, om, gm = _models(name, torch)
    x, y = cases.eval_point_for(name, om)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    ref = (gm.obj(xd), gm.grad(xd).clone(), gm.cons(xd).clone(), gm.jac_coord(xd).clone(), gm.hess_coord(xd, yd, obj_weight=0.7).clone())
    for kw in (dict(fused=False, defer_obj=False), dict(fused=False, defer_obj=True), dict(fused=True, defer_obj=True)):
        nan = lambda n: torch.full((n,), float("nan"), device="cuda", dtype=torch.float64)
        g, c, j, h = nan(om.nvar), nan(om.ncon), nan(om.nnzj), nan(om.nnzh)
        step = gm.raw_loop(xd, yd, g, c, j, h, obj_weight=0.7, **kw)
        for _ in range(3):
            f = step()
        torch.cuda.synchronize()
        assert f == ref[0] and torch.equal(g, ref[1]) and torch.equal(c, ref[2]) and torch.equal(j, ref[3]) and torch.equal(h, ref[4]), kw
    gm.close()


@pytest.mark.parametrize("name,builder", [("quadrotor_oc3", lambda: __import__("infiniteexamodels.jl_amd.workloads", fromlist=["x"]).quadrotor(40_000, collocation=3)),
                                          ("kinetic", lambda: __import__("infiniteexamodels.jl_amd.workloads", fromlist=["x"]).kinetic_control(30_000)),
                                          ("hovercraft_oc4", lambda: __import__("infiniteexamodels.jl_amd.workloads", fromlist=["x"]).hovercraft(20_001, collocation=4)),
                                          ("pandemic_oc3", lambda: __import__("infiniteexamodels.jl_amd.workloads", fromlist=["x"]).pandemic(3_000, 12, collocation=3))])
def test_collocation_fold_variants_on_gpu(name, builder, torch_cuda):
    """`fold_colloc` on grids of > 64 workgroups (the lane-fused shape): 1 (default) = the derivative rows of an
    orthogonal-collocation model ride on the support lanes for grad! / jtprod! / hprod! (exclusive stores, no gather
    plan), 2 = for every kind, 0 = the plan-driven gather.  Every variant agrees with the oracle; the COO kinds write the
    same bytes in all three; jtprod! is bitwise reproducible from call to call."""
    torch = torch_cuda
    from infiniteexamodels.jl_amd import transcribe
    from infiniteexamodels.jl_amd.model import ExaModel
    from pyoracle import OracleModel
    core = transcribe.exa_core(builder())
    blob = core.to_blob()
    om = OracleModel(blob)
    x, y = cases.eval_point_for(name, om, 3)
    rng = np.random.default_rng(11)
    v, vc = rng.standard_normal(om.nvar), rng.standard_normal(om.ncon)
    xd, yd, vd, vcd = (torch.tensor(a, device="cuda") for a in (x, y, v, vc))
    nanv = lambda n: torch.full((n,), float("nan"), device="cuda", dtype=torch.float64)
    ref = dict(cons=om.cons(x), jac=om.jac_coord(x), jprod=om.jprod(x, v), jtprod=om.jtprod(x, vc), grad=om.grad(x), hprod=om.hprod(x, y, v, 0.3))
    coo = {}
    for fc in (1, 2, 0):
        gm = ExaModel(core, device=0, blob=blob, options=dict(fold_colloc=fc))
        out = dict(cons=gm.cons(xd, nanv(om.ncon)), jac=gm.jac_coord(xd, nanv(om.nnzj)), jprod=gm.jprod(xd, vd, nanv(om.ncon)),
                   jtprod=gm.jtprod(xd, vcd, nanv(om.nvar)), grad=gm.grad(xd, nanv(om.nvar)), hprod=gm.hprod(xd, yd, vd, nanv(om.nvar), obj_weight=0.3))
        for k, t in out.items():
            _close(t.cpu().numpy(), ref[k], f"{k} (fold_colloc={fc})")
        hv = gm.hess_coord(xd, yd, nanv(om.nnzh), obj_weight=0.3)
        _close(hv.cpu().numpy(), om.hess_coord(x, y, 0.3), f"hess_coord (fold_colloc={fc})")
        assert abs(gm.obj(xd) - om.obj(x)) <= RTOL * max(1.0, abs(om.obj(x)))
        coo[fc] = (out["cons"].clone(), out["jac"].clone(), hv.clone())
        first = out["jtprod"].clone()
        for _ in range(5):
            assert torch.equal(gm.jtprod(xd, vcd, nanv(om.nvar)), first), f"jtprod! not reproducible (fold_colloc={fc})"
        gm.close()
    for fc in (2, 0):
        for a, b in zip(coo[1], coo[fc]):
            assert torch.equal(a, b), f"cons! / jac_coord! / hess_coord! bytes differ between fold_colloc=1 and {fc}"
