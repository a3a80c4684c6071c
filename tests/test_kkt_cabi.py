"""The chain KKT solver as one object behind the C-ABI (iem_kkt_create / _assemble / _factor / _solve; host analysis in
csrc/iem_kkt_host.hpp).  CPU: the C++ analysis against the Python one it was ported from (kkt_chain.ChainLayout) — the
grouping field for field, and the gather plan replayed in numpy against the dense blocks the Python scatter plan fills.
GPU: assemble / factor / solve through the C-ABI against scipy on the oracle's KKT matrix."""
import ctypes as C

import numpy as np
import pytest
from scipy.sparse.linalg import spsolve

import cases
import chain_reference as ref
from pyoracle import OracleModel
from test_kkt import host_kkt

MODELS = ["quadrotor_100", "quadrotor_5", "quadrotor_oc3_40", "farmer_5", "opf_7", "pandemic_20x3", "hovercraft", "hovercraft_oc4", "kinetic_20", "test_problem_1", "pandemic_100x7"]


@pytest.mark.parametrize("name", MODELS)
def test_host_analysis_matches_the_python_layout(name, built):
    from infiniteexamodels.jl_amd import lib as iemlib
    from infiniteexamodels.jl_amd.kkt_chain import ChainLayout
    core = cases.build_core(name)
    blob = core.to_blob()
    om = OracleModel(blob)
    jr, jc = om.jac_structure()
    L = ChainLayout(core.slabs, om.nvar, om.ncon, jr, jc)
    info, blk, loc, rows, cols, dest, seg, perm = iemlib.kkt_analyse_blob(blob)
    assert (info["S"], info["nb"], info["ne"], info["nc"], info["reach"], info["group"], info["phase"], info["n_border"]) == \
           (L.S, L.nb, L.ne, L.nc, L.reach, L.group, L.phase, L.n_border)
    assert np.array_equal(blk, L.blk) and np.array_equal(loc, L.loc)
    rt, ct = L.coupling_tables()
    assert np.array_equal(rows, rt) and np.array_equal(cols, ct)
    # the gather plan, replayed: blocks from the COO values == the blocks the Python scatter plan fills from the CSR matrix
    x, y = cases.eval_point_for(name, om, 5)
    sigma = 0.5 + np.random.default_rng(3).random(om.nvar)
    dw, dc = 1e-2, 1e-6
    src = np.concatenate([om.hess_coord(x, y, 1.0), om.jac_coord(x), sigma + dw, np.full(om.ncon, -dc), [1.0]])
    flat = np.zeros(info["block_doubles"])
    vals = np.add.reduceat(src[perm], seg[:-1].astype(np.int64)) if len(perm) else np.zeros(0)
    assert np.unique(dest).size == dest.size
    flat[dest] = vals
    K = host_kkt(om, x, y, sigma, dw, dc).tocsr()
    K.sum_duplicates(); K.sort_indices()
    n = om.nvar + om.ncon
    D, B, E, G = ref.fill_blocks(L, np.repeat(np.arange(n), np.diff(K.indptr)), K.indices, K.data)
    oD, oB, oE, oG, total = L.offsets()
    assert total == info["block_doubles"]
    assert np.allclose(flat[oD:oB].reshape(L.S, L.nb, L.nb), D, rtol=0, atol=1e-12 * max(1.0, np.abs(D).max()))
    if L.reach > 0:
        Bt = flat[oB:oE].reshape(L.S, L.nc, L.nc)
        assert np.allclose(Bt[:, :L.rowsR.size, :L.colsC.size], B[:, L.rowsR[:, None], L.colsC[None, :]], rtol=0, atol=1e-12)
    assert np.allclose(flat[oE:oG].reshape(L.S, L.nb, L.ne), E, rtol=0, atol=1e-12 * max(1.0, np.abs(E).max() if E.size else 1.0))
    assert np.allclose(flat[oG:total].reshape(L.ne, L.ne), G, rtol=0, atol=1e-12 * max(1.0, np.abs(G).max() if G.size else 1.0))


@pytest.mark.parametrize("name", ["pandemic_300x7", "pandemic_200x24", "pandemic2_150x8"])      # (the last: two hubs per time block)
def test_host_analysis_of_a_hub_border_matches_the_python_one(name, built):
    """A border beyond 128 unknowns on a laned grid: iem_kkt_create keeps it as span-sparse hubs.  The C++ analysis (grouping, hub
    order, the rows Q, the gather plan into D | Bt | E0 | S) against kkt_chain.HubChainKKT's, entry for entry."""
    import types
    import torch
    from infiniteexamodels.jl_amd import lib as iemlib
    from infiniteexamodels.jl_amd.kkt_chain import HubChainKKT
    core = cases.build_core(name)
    blob = core.to_blob()
    om = OracleModel(blob)
    info, blk, loc, rows, cols, dest, seg, perm = iemlib.kkt_analyse_blob(blob)
    x, y = cases.eval_point_for(name, om, 5)
    sigma = 0.5 + np.random.default_rng(3).random(om.nvar)
    dw, dc = 1e-2, 1e-6
    K = host_kkt(om, x, y, sigma, dw, dc).tocsr()
    K.sum_duplicates(); K.sort_indices()
    n = om.nvar + om.ncon
    stub = types.SimpleNamespace(core=core, meta=types.SimpleNamespace(nvar=om.nvar, ncon=om.ncon), jac_structure=lambda base=0: om.jac_structure(), device="cpu")
    kkt = types.SimpleNamespace(model=stub, n=n, rowptr=torch.as_tensor(K.indptr.astype(np.int32)), colind=torch.as_tensor(K.indices.astype(np.int32)), vals=torch.as_tensor(K.data))
    hub = HubChainKKT(kkt, levels=ref.HubLevels(), device="cpu").load()
    L = hub.layout
    assert info["hubs"] == 1 and info["ne"] == 0 and (info["S"], info["lanes"], info["nb"], info["nc"], info["n_border"]) == (L.S, L.lanes, L.nb, L.nc, L.n_border)
    assert (info["hub_rows"], info["hubs_per_block"], info["hub_ld"]) == (hub.nQ, hub.hw, hub.Hp)
    chain = L.blk >= 0
    assert np.array_equal(blk, L.blk) and np.array_equal(loc[chain], L.loc[chain])
    src = np.concatenate([om.hess_coord(x, y, 1.0), om.jac_coord(x), sigma + dw, np.full(om.ncon, -dc), [1.0]])
    flat = np.zeros(info["block_doubles"])
    assert np.unique(dest).size == dest.size
    flat[dest] = np.add.reduceat(src[perm], seg[:-1].astype(np.int64))
    want = np.concatenate([hub.flat.numpy(), hub.E0.numpy(), hub.Sbig.numpy()])
    assert flat.size == want.size and np.abs(flat - want).max() <= 1e-12 * np.abs(want).max()


def test_host_analysis_agrees_on_every_small_case(built):
    """grouping, coupling width, block phase — or the refusal — for every model of tests/cases.py"""
    import warnings
    from infiniteexamodels.jl_amd import lib as iemlib
    from infiniteexamodels.jl_amd.kkt_chain import ChainLayout
    for name in cases.small_cases():
        if name in ("quadrotor_1000", "farmer_1000", "opf_600", "pandemic_300x7", "quadrotor_oc3_700"):
            continue
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            core = cases.build_core(name)
        blob = core.to_blob()
        om = OracleModel(blob)
        jr, jc = om.jac_structure()
        try:
            L = ChainLayout(core.slabs, om.nvar, om.ncon, jr, jc)
            py = (L.S, L.nb, L.ne, L.nc, L.reach, L.group, L.phase, L.n_border)
        except iemlib.IemError as e:
            py = str(e).split(":", 1)[1][:50]
        try:
            info = iemlib.kkt_analyse_blob(blob)[0]
            cc = tuple(info[k] for k in ("S", "nb", "ne", "nc", "reach", "group", "phase", "n_border"))
        except iemlib.IemError as e:
            cc = str(e).split("chain KKT:", 1)[1][:50] if "chain KKT:" in str(e) else str(e)
        if isinstance(py, str):
            assert isinstance(cc, str) and py.split("(")[0].strip()[:24] in cc, (name, py, cc)      # both refuse, for the same reason
        else:
            assert py == cc, (name, py, cc)


def test_analysis_refuses_what_the_solver_cannot_hold(built):
    from infiniteexamodels.jl_amd import lib as iemlib, transcribe, workloads
    big = transcribe.exa_core(workloads.pandemic(290, 40)).to_blob()     # 40 scenarios x 17 unknowns per time support; as lanes: a border of 300 —
    info = iemlib.kkt_analyse_blob(big)[0]                                # beyond the dense-border kernels (128): kept as span-sparse HUBS, ne = 0 for the kernels
    assert (info["hubs"], info["lanes"], info["S"], info["nb"], info["ne"], info["n_border"], info["hubs_per_block"], info["hub_ld"]) == (1, 40, 40 * 300, 20, 0, 300, 1, 512)
    # ... while the same model on the reference's ladder grid (ESCAPE34/run_cases_gpu.jl:99-102: 100 + 10 supports) runs as lanes
    info = iemlib.kkt_analyse_blob(transcribe.exa_core(workloads.pandemic(100, 40)).to_blob())[0]
    assert (info["S"], info["nb"], info["ne"], info["n_border"], info["nc"]) == (40 * 110, 20, 112, 110, 4)
    with pytest.raises(iemlib.IemError, match="no infinite-parameter slab table"):   # a hand-built core: nothing to chain along
        iemlib.kkt_analyse_blob(cases.build_core("wide_rows").to_blob())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["quadrotor_100", "quadrotor_oc3_40", "farmer_5", "opf_7", "pandemic_20x3", "hovercraft", "kinetic_20", "quadrotor_1000", "opf_600", "farmer_1000",
                                  "pandemic_100x7", "pandemic_300x7", "pandemic_200x24", "pandemic2_150x8"])      # (the last three: 200 - 310 border unknowns -> HUB mode)
def test_assemble_factor_solve_through_the_c_abi(name, built):
    import torch
    from infiniteexamodels.jl_amd import lib as iemlib
    from infiniteexamodels.jl_amd.model import ExaModel
    core = cases.build_core(name)
    blob = core.to_blob()
    om = OracleModel(blob)
    gm = ExaModel(core, device=0, blob=blob)
    L_ = gm._L
    k = C.c_void_p()
    iemlib.check(L_.iem_kkt_create(gm._h, 0, C.byref(k)))
    info = iemlib.KktInfo()
    iemlib.check(L_.iem_kkt_info(k, C.byref(info)))
    n = om.nvar + om.ncon
    assert info.n == n and bool(info.hubs) == (name in ("pandemic_300x7", "pandemic_200x24", "pandemic2_150x8"))
    rng = np.random.default_rng(3)
    p = lambda t: C.c_void_p(t.data_ptr())
    for seed in (5, 9):
        x, y = cases.eval_point_for(name, om, seed)
        sigma = 0.5 + rng.random(om.nvar)
        xd, yd, sd = (torch.tensor(a, device="cuda") for a in (x, y, sigma))
        hv, jv = gm.hess_coord(xd, yd, obj_weight=1.0), gm.jac_coord(xd)
        gm._sync_stream()
        iemlib.check(L_.iem_kkt_assemble(k, p(hv), p(jv), p(sd), 1e-2, 1e-6))
        inertia = (C.c_int64 * 3)()
        iemlib.check(L_.iem_kkt_factor(k, inertia))
        Kh = host_kkt(om, x, y, sigma, 1e-2, 1e-6)
        neg_ref = int((np.linalg.eigvalsh(Kh.toarray()) < 0).sum()) if n <= 4000 else None
        if neg_ref is not None:
            assert (inertia[0], inertia[1], inertia[2]) == (n - neg_ref, neg_ref, 0)
        if info.hubs:            # too large for dense eigenvalues: the Python-held form of the same pipeline (checked against them at small sizes) as the witness
            from infiniteexamodels.jl_amd.kkt import KKTSystem
            from infiniteexamodels.jl_amd.kkt_chain import HubChainKKT
            kk = KKTSystem(gm)
            kk.assemble(hv, jv, sd, 1e-2, 1e-6)
            assert tuple(inertia) == HubChainKKT(kk).load().factor().inertia() and inertia[1] >= om.ncon and inertia[2] == 0
            kk.close()
        rhs = rng.standard_normal(n)
        rd, sol = torch.tensor(rhs, device="cuda"), torch.empty(n, dtype=torch.float64, device="cuda")
        iemlib.check(L_.iem_kkt_solve(k, p(rd), p(sol)))
        xs = sol.cpu().numpy()
        res = rhs - Kh @ xs                      # one step of refinement, the residual formed on the host here
        rd2 = torch.tensor(res, device="cuda")
        iemlib.check(L_.iem_kkt_solve(k, p(rd2), p(rd2)))     # in place: the right-hand side is read before the solution is written
        xs = xs + rd2.cpu().numpy()
        want = spsolve(Kh.tocsc(), rhs)
        resid = np.abs(Kh @ xs - rhs)
        assert resid.max() <= 1e-9 * max(1.0, np.abs(rhs).max()) or (resid / (abs(Kh) @ np.abs(xs) + np.abs(rhs))).max() <= 1e-12, name
        np.testing.assert_allclose(xs, want, rtol=1e-6, atol=1e-8 * max(1.0, np.abs(want).max()))
    iemlib.check(L_.iem_kkt_destroy(k))
    gm.close()


@pytest.mark.gpu
def test_a_singular_system_is_reported_as_doubtful(built):
    """[0 J'; J 0] of the farmer LP without any regularisation (sigma = 0, delta_w = delta_c = 0) is singular (nvar > ncon): a host
    doing the usual inertia correction (neg == ncon and doubtful == 0) must see DOUBTFUL pivots — from the blocks or from the
    border's Schur complement, whose zero eigenvalues count as doubtful, never as positive — and shift."""
    import torch
    from infiniteexamodels.jl_amd import lib as iemlib
    from infiniteexamodels.jl_amd.model import ExaModel
    core = cases.build_core("farmer_5")
    blob = core.to_blob()
    om = OracleModel(blob)
    gm = ExaModel(core, device=0, blob=blob)
    k = C.c_void_p()
    iemlib.check(gm._L.iem_kkt_create(gm._h, 0, C.byref(k)))
    x, y = cases.eval_point_for("farmer_5", om, 5)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    sd = torch.zeros(om.nvar, dtype=torch.float64, device="cuda")
    hv, jv = gm.hess_coord(xd, yd, obj_weight=1.0), gm.jac_coord(xd)
    gm._sync_stream()
    p = lambda t: C.c_void_p(t.data_ptr())
    iemlib.check(gm._L.iem_kkt_assemble(k, p(hv), p(jv), p(sd), 0.0, 0.0))
    inertia = (C.c_int64 * 3)()
    iemlib.check(gm._L.iem_kkt_factor(k, inertia))
    assert inertia[2] > 0, tuple(inertia)
    # regularised, the same object factorises cleanly again
    iemlib.check(gm._L.iem_kkt_assemble(k, p(hv), p(jv), p(sd), 1e-2, 1e-6))
    iemlib.check(gm._L.iem_kkt_factor(k, inertia))
    assert (inertia[1], inertia[2]) == (om.ncon, 0), tuple(inertia)
    iemlib.check(gm._L.iem_kkt_destroy(k))
    gm.close()


@pytest.mark.gpu
def test_a_singular_system_is_reported_as_doubtful_in_hub_mode(built):
    """The same for a hub border (pandemic 310 x 7): without any regularisation the KKT matrix of the point is singular or
    numerically so — the chain's blocks or the hubs' pivot blocks (threshold relative to the largest diagonal entry of their Schur
    complement) must report DOUBTFUL pivots; regularised, the same object factorises cleanly again."""
    import torch
    from infiniteexamodels.jl_amd import lib as iemlib
    from infiniteexamodels.jl_amd.model import ExaModel
    core = cases.build_core("pandemic_300x7")
    blob = core.to_blob()
    om = OracleModel(blob)
    gm = ExaModel(core, device=0, blob=blob)
    k = C.c_void_p()
    iemlib.check(gm._L.iem_kkt_create(gm._h, 0, C.byref(k)))
    x, y = cases.eval_point_for("pandemic_300x7", om, 5)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    sd = torch.zeros(om.nvar, dtype=torch.float64, device="cuda")
    hv, jv = gm.hess_coord(xd, yd, obj_weight=1.0), gm.jac_coord(xd)
    gm._sync_stream()
    p = lambda t: C.c_void_p(t.data_ptr())
    iemlib.check(gm._L.iem_kkt_assemble(k, p(hv), p(jv), p(sd), 0.0, 0.0))
    inertia = (C.c_int64 * 3)()
    iemlib.check(gm._L.iem_kkt_factor(k, inertia))
    assert inertia[2] > 0, tuple(inertia)
    iemlib.check(gm._L.iem_kkt_assemble(k, p(hv), p(jv), p(sd + 0.5), 1e-2, 1e-6))
    iemlib.check(gm._L.iem_kkt_factor(k, inertia))
    assert inertia[1] >= om.ncon and inertia[2] == 0 and inertia[0] + inertia[1] == om.nvar + om.ncon, tuple(inertia)
    iemlib.check(gm._L.iem_kkt_destroy(k))
    gm.close()
