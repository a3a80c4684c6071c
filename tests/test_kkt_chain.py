"""Chain KKT solver.  CPU: the grouping of the unknowns into chain blocks + border derived from the slab
table and the Jacobian structure (kkt_chain.ChainLayout), its scatter plan, and the block cyclic reduction itself (numpy
restatement, tests/chain_reference.py) against scipy on the oracle's KKT matrix.  GPU: the hand-written kernels
(csrc/iem_kkt_device.h through iem_kkt_chain_factor / _solve) against the same."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.sparse.linalg import spsolve

import cases
import chain_reference as ref
from pyoracle import OracleModel
from test_kkt import host_kkt

MODELS = ["quadrotor_100", "quadrotor_5", "quadrotor_oc3_40", "farmer_5", "opf_7", "pandemic_20x3", "hovercraft", "test_problem_1", "kinetic_20", "irregular", "pandemic_100x7"]


def _system(name, seed=5):
    core = cases.build_core(name)
    om = OracleModel(core.to_blob())
    x, y = cases.eval_point_for(name, om, seed)
    rng = np.random.default_rng(3)
    sigma = 0.5 + rng.random(om.nvar)
    K = host_kkt(om, x, y, sigma, 1e-2, 1e-6).tocsr()
    K.sum_duplicates(); K.sort_indices()
    return core, om, K, rng.standard_normal(om.nvar + om.ncon)


@pytest.mark.parametrize("name", MODELS)
def test_layout_and_block_cyclic_reduction_on_cpu(name, built):
    from infiniteexamodels.jl_amd.kkt_chain import ChainLayout
    core, om, K, rhs = _system(name)
    jr, jc = om.jac_structure()
    L = ChainLayout(core.slabs, om.nvar, om.ncon, jr, jc)
    n = om.nvar + om.ncon
    assert L.counts.sum() + L.n_border == n and L.counts.max() <= L.nb and L.nb % 4 == 0 and L.ne % 4 == 0
    # every unknown has one slot; variables precede rows inside a block
    slot = np.where(L.blk >= 0, L.blk * L.nb + L.loc, -1 - L.loc)
    assert np.unique(slot).size == n
    rows = np.repeat(np.arange(n), np.diff(K.indptr))
    D, B, E, G = ref.fill_blocks(L, rows, K.indices, K.data)
    Dinv, X, Y, Z, Gp, neg = ref.factor(D, B, E)
    on, pos, border = L.positions()

    def solve(b):
        r = np.zeros(L.S * L.nb); r[pos] = b[on]
        rB = np.zeros(L.ne); rB[:L.n_border] = b[border]
        xs, xB = ref.solve(Dinv, X, Y, Z, G, Gp, r.reshape(L.S, L.nb), rB)
        out = np.empty(n); out[on] = xs.reshape(-1)[pos]; out[border] = xB[:L.n_border]
        return out
    sol = solve(rhs)
    assert np.abs(K @ sol - rhs).max() <= 1e-3 * max(1.0, np.abs(rhs).max())     # pivot blocks carry the -delta_c rows: ~1e-4 before ...
    sol = sol + solve(rhs - K @ sol)                                               # ... one step of iterative refinement (ChainKKT.solve's default)
    want = spsolve(K.tocsc(), rhs)
    assert np.abs(K @ sol - rhs).max() <= 1e-8 * max(1.0, np.abs(rhs).max())
    np.testing.assert_allclose(sol, want, rtol=1e-6, atol=1e-8 * max(1.0, np.abs(want).max()))
    # inertia: the negative pivots (blocks + border) are the negative eigenvalues of K — ncon of them when the
    # regularised Hessian block is positive definite, more when it is not (random multipliers: the quadrotor's is not),
    # which is exactly what an interior-point method asks the factorisation for
    Gs = G - Gp.sum(0)
    assert neg + int((np.linalg.eigvalsh(Gs) < 0).sum()) == int((np.linalg.eigvalsh(K.toarray()) < 0).sum()) >= om.ncon


def test_layout_picks_the_stencil_axis_and_refuses_oversized_blocks(built):
    from infiniteexamodels.jl_amd import lib as iemlib, transcribe, workloads
    from infiniteexamodels.jl_amd.kkt_chain import ChainLayout
    core = cases.build_core("pandemic_20x3")                 # groups: t (stencil) and xi — the chain must run along t
    om = OracleModel(core.to_blob())
    L = ChainLayout(core.slabs, om.nvar, om.ncon, *om.jac_structure())
    assert L.group == 1 and L.reach == 1 and L.S == 30 and L.n_border == 0      # 20 + 10 extra supports
    q = cases.build_core("quadrotor_oc3_40")                 # OrthogonalCollocation(3): an element's rows reach two supports back
    oq = OracleModel(q.to_blob())
    Lq = ChainLayout(q.slabs, oq.nvar, oq.ncon, *oq.jac_structure())
    assert Lq.reach == 2 and Lq.supports_per_block == 2
    f = cases.build_core("farmer_5")                         # no stencil: independent scenario blocks + first-stage border
    of = OracleModel(f.to_blob())
    Lf = ChainLayout(f.slabs, of.nvar, of.ncon, *of.jac_structure())
    assert Lf.reach == 0 and Lf.S == 5 and Lf.n_border == 4   # x[1:3] and the row sum(x) <= 500
    wide = transcribe.exa_core(workloads.pandemic(10, 40))    # 40 scenarios x 17 unknowns per time support: too wide for dense blocks —
    ow = OracleModel(wide.to_blob())                          # one chain per scenario (LANES), u(t) in the border
    Lw = ChainLayout(wide.slabs, ow.nvar, ow.ncon, *ow.jac_structure())
    assert (Lw.lanes, Lw.S, Lw.nb, Lw.n_border, Lw.nc, Lw.reach) == (40, 40 * 20, 20, 20, 4, 1)
    assert (Lw.blk[Lw.blk >= 0] // 20).max() == 39 and set(np.nonzero(Lw.blk[:ow.nvar] < 0)[0]) == set(range(4 * 20 * 40, 4 * 20 * 40 + 20))   # the border IS the slab of u(t)
    big = transcribe.exa_core(workloads.pandemic(290, 40))    # ... and a border of 300 time supports is too large for that too
    ob = OracleModel(big.to_blob())
    with pytest.raises(iemlib.IemError, match="exceed the dense-block solver's limits.*one chain per lane"):
        ChainLayout(big.slabs, ob.nvar, ob.ncon, *ob.jac_structure())


def _pandemic_system(nt, nxi):
    from infiniteexamodels.jl_amd import transcribe, workloads
    core = transcribe.exa_core(workloads.pandemic(nt, nxi) if nxi > 0 else cases.pandemic_two_controls(nt, -nxi))   # (nxi < 0: the variant with two controls on t)
    om = OracleModel(core.to_blob())
    x, y = cases.eval_point_for("pandemic_x", om, 5)
    rng = np.random.default_rng(3)
    sigma = 0.5 + rng.random(om.nvar)
    K = host_kkt(om, x, y, sigma, 1e-2, 1e-6).tocsr()
    K.sum_duplicates(); K.sort_indices()
    return core, om, (x, y, sigma), K, rng.standard_normal(om.nvar + om.ncon)


@pytest.mark.parametrize("nt,nxi", [(10, 4), (23, 3), (54, 5), (13, -3)])      # 20 / 33 / 64 / 23 time blocks per lane; the last with TWO hubs per time block
def test_hub_border_pipeline_on_cpu(nt, nxi, built):
    """kkt_chain.HubChainKKT (config 3's solver: one chain per scenario, u(t) as span-sparse hubs) with the chain levels done
    by the dense restatement (chain_reference.HubLevels) instead of the device: the span bookkeeping of the hubs' columns, the
    accumulation of their Schur complement, its block LDL' and the two-chain-solve substitution against scipy's sparse LU, and
    the inertia against the eigenvalues."""
    import types
    import torch
    from infiniteexamodels.jl_amd.kkt_chain import HubChainKKT
    core, om, _, K, rhs = _pandemic_system(nt, nxi)
    n = om.nvar + om.ncon
    stub = types.SimpleNamespace(core=core, meta=types.SimpleNamespace(nvar=om.nvar, ncon=om.ncon), jac_structure=lambda base=0: om.jac_structure(), device="cpu")
    kkt = types.SimpleNamespace(model=stub, n=n, rowptr=torch.as_tensor(K.indptr.astype(np.int32)), colind=torch.as_tensor(K.indices.astype(np.int32)),
                                vals=torch.as_tensor(K.data))
    hub = HubChainKKT(kkt, levels=ref.HubLevels(), device="cpu")
    assert hub.lanes == abs(nxi) and hub.hw == (1 if nxi > 0 else 2) and hub.H == hub.hw * (nt + 10) and hub.nb == 20 and hub.Tp == nt + 10
    hub.load().factor()
    neg_ref = int((np.linalg.eigvalsh(K.toarray()) < 0).sum())
    assert hub.inertia() == (n - neg_ref, neg_ref, 0)
    sol = hub.solve(torch.as_tensor(rhs)).numpy()
    sol = sol + hub.solve(torch.as_tensor(rhs - K @ sol)).numpy()
    want = spsolve(K.tocsc(), rhs)
    assert np.abs(K @ sol - rhs).max() <= 1e-9 * max(1.0, np.abs(rhs).max())
    np.testing.assert_allclose(sol, want, rtol=1e-7, atol=1e-9 * np.abs(want).max())


@pytest.mark.gpu
@pytest.mark.parametrize("nt,nxi", [(54, 5), (190, 24), (100, 7), (140, -8)])    # 64 / 200 / 110 / 150 time blocks per lane (200 hubs = 3 pivot blocks); the last with two hubs per time block
def test_hub_border_chain_kkt_on_gpu(nt, nxi, built):
    """The same through the device: iem_kkt_chain_level / iem_kkt_chain_solve for the lanes' chains, kkt_eliminate for the hubs'
    pivot blocks (the C-ABI entries), torch for the span-sparse GEMMs between them."""
    import torch
    from infiniteexamodels.jl_amd.kkt import KKTSystem
    from infiniteexamodels.jl_amd.kkt_chain import HubChainKKT
    from infiniteexamodels.jl_amd.model import ExaModel
    core, om, (x, y, sigma), K, rhs = _pandemic_system(nt, nxi)
    n = om.nvar + om.ncon
    gm = ExaModel(core, device=0)
    kkt = KKTSystem(gm)
    hub = HubChainKKT(kkt)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    for _ in range(2):                                                 # (twice: load() must restore everything factor() overwrote)
        kkt.assemble(gm.hess_coord(xd, yd, obj_weight=1.0), gm.jac_coord(xd), torch.tensor(sigma, device="cuda"), 1e-2, 1e-6)
        hub.load().factor()
        pos, neg, doubtful = hub.inertia()
        assert pos + neg == n and doubtful == 0 and neg >= om.ncon
        if n <= 4000:
            assert neg == int((np.linalg.eigvalsh(K.toarray()) < 0).sum())
        r = torch.tensor(rhs, device="cuda")
        sol = hub.solve(r).cpu().numpy()
        sol = sol + hub.solve(torch.tensor(rhs - K @ sol, device="cuda")).cpu().numpy()
        want = spsolve(K.tocsc(), rhs)
        assert np.abs(K @ sol - rhs).max() <= 1e-9 * max(1.0, np.abs(rhs).max())
        np.testing.assert_allclose(sol, want, rtol=1e-6, atol=1e-8 * np.abs(want).max())
    kkt.close(); gm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nt,nxi", [(54, 5), (30, -4)])
def test_hub_level_kernels_against_the_library_products(nt, nxi, built):
    """iem_kkt_hub_level (kkt_hub_z / kkt_hub_widen) on its own: Z = D^-1[Q, Q] E of the eliminated blocks and the survivors' widened
    columns, level by level, against the same quantities formed with gathered operands and batched library products."""
    import torch
    from infiniteexamodels.jl_amd.kkt import KKTSystem
    from infiniteexamodels.jl_amd.kkt_chain import HubChainKKT
    from infiniteexamodels.jl_amd.model import ExaModel
    core, om, (x, y, sigma), K, rhs = _pandemic_system(nt, nxi)
    gm = ExaModel(core, device=0)
    kkt = KKTSystem(gm)
    hub = HubChainKKT(kkt)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    kkt.assemble(gm.hess_coord(xd, yd, obj_weight=1.0), gm.jac_coord(xd), torch.tensor(sigma, device="cuda"), 1e-2, 1e-6)
    hub.load()
    gm._sync_stream()
    t = torch
    lanes, Tp, nb, nc, hw, nQ = hub.lanes, hub.Tp, hub.nb, hub.nc, hub.hw, hub.nQ
    D4, Bt = hub.D.view(lanes, Tp, nb * nb), hub.Bt.view(lanes, Tp, nc, nc)
    R, Cc, Q = hub._qR, hub._qC, hub._Q
    nR, nC = int(R.numel()), int(Cc.numel())
    qq = (Q[:, None] * nb + Q[None, :]).reshape(-1)
    E = hub.E0.view(Tp, lanes, nQ, hw).clone()
    E += 0.01 * t.randn_like(E)                      # (fill every row of Q: at the first levels most of E is structurally zero)
    hub._level(1, 3)
    s, levels = 1, 0
    while s < Tp:
        hub._level(s, 0)
        W = (2 * s - 1) * hw
        n_alive = E.shape[0]; n_e, n_s = n_alive // 2, n_alive - n_alive // 2
        Z = t.empty(n_e, lanes, nQ, W, dtype=t.float64, device="cuda")
        En = t.empty(n_s, lanes, nQ, (4 * s - 1) * hw, dtype=t.float64, device="cuda")
        hub._hub_level(s, E, Z, En, 0)
        Dq = D4[:, s::2 * s][:, :, qq].permute(1, 0, 2).reshape(n_e, lanes, nQ, nQ)
        Zr = t.matmul(Dq, E[1::2])
        Er = t.zeros_like(En)
        Er[..., s * hw: s * hw + W] = E[0::2]
        if n_s > 1:
            Er[1:, :, R, 0:W] -= t.matmul(Bt[:, 2 * s::2 * s].permute(1, 0, 2, 3)[:, :, :nR, :nC], Zr[:n_s - 1][:, :, Cc, :])
        Er[:n_e, :, Cc, 2 * s * hw: 2 * s * hw + W] -= t.matmul(Bt[:, s::2 * s].permute(1, 0, 2, 3)[:, :, :nR, :nC].transpose(-1, -2), Zr[:, :, R, :])
        scale = max(1.0, float(Zr.abs().max()))
        assert float((Z - Zr).abs().max()) <= 1e-12 * scale and float((En - Er).abs().max()) <= 1e-12 * max(scale, float(Er.abs().max())), s
        E = En
        hub._level(s, 1)
        s *= 2; levels += 1
    assert levels >= 5 and E.shape[0] == 1
    kkt.close(); gm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", MODELS + ["quadrotor_1000", "opf_600", "farmer_1000"])
def test_chain_kkt_on_gpu(name, built):
    import torch
    from infiniteexamodels.jl_amd.kkt import KKTSystem
    from infiniteexamodels.jl_amd.kkt_chain import ChainKKT
    from infiniteexamodels.jl_amd.model import ExaModel
    core = cases.build_core(name)
    blob = core.to_blob()
    om = OracleModel(blob)
    gm = ExaModel(core, device=0, blob=blob)
    kkt = KKTSystem(gm)
    ck = ChainKKT(kkt)
    rng = np.random.default_rng(3)
    n = om.nvar + om.ncon
    for seed in (5, 9):
        x, y = cases.eval_point_for(name, om, seed)
        sigma = 0.5 + rng.random(om.nvar)
        xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
        kkt.assemble(gm.hess_coord(xd, yd, obj_weight=1.0), gm.jac_coord(xd), torch.tensor(sigma, device="cuda"), 1e-2, 1e-6)
        Kh = host_kkt(om, x, y, sigma, 1e-2, 1e-6)
        ck.load().factor()
        pos, neg, doubtful = ck.inertia()
        # the factors themselves, block for block, and the pivot signs, against the numpy restatement
        L = ck.layout
        Kc = Kh.tocsr(); Kc.sum_duplicates(); Kc.sort_indices()
        D, B, E, G = ref.fill_blocks(L, np.repeat(np.arange(n), np.diff(Kc.indptr)), Kc.indices, Kc.data)
        Dinv, X, Y, Z, Gp, neg_ref = ref.factor(D, B, E)
        neg_ref += int((np.linalg.eigvalsh(G - Gp.sum(0)) < 0).sum())
        assert (pos, neg, doubtful) == (n - neg_ref, neg_ref, 0) and neg >= om.ncon
        got = ck.D.view(L.S, L.nb, L.nb).cpu().numpy()
        err = np.abs(got - Dinv).max() / max(1.0, np.abs(Dinv).max())
        print(f"{name}: max |Dinv - ref| / max |ref| = {err:.2e}  (max |ref| {np.abs(Dinv).max():.2e})")
        assert err <= 2e-5      # (unpivoted block Gauss-Jordan here, pivoted LU there; blocks with -delta_c pivots reach 1e6)
        rhs = rng.standard_normal(n)
        sol = ck.solve(torch.tensor(rhs, device="cuda"), refine=1).cpu().numpy()
        want = spsolve(Kh.tocsc(), rhs)
        resid = np.abs(Kh @ sol - rhs)
        # 1e-9 of the right-hand side, or — where the solution itself is 1e5..1e6 (farmer: multipliers over delta_c) — a
        # componentwise backward error at rounding level
        assert resid.max() <= 1e-9 * max(1.0, np.abs(rhs).max()) or (resid / (abs(Kh) @ np.abs(sol) + np.abs(rhs))).max() <= 1e-12, name
        np.testing.assert_allclose(sol, want, rtol=1e-6, atol=1e-8 * max(1.0, np.abs(want).max()))
    kkt.close(); gm.close()


@pytest.mark.gpu
def test_device_resident_newton_iterations_reach_a_kkt_point(built):
    """f3 + f4 together: Lagrange-Newton iterations whose every step — the five evaluation calls, the KKT assembly, the chain
    factorisation and solve — runs on the device (tools/newton_kkt_demo.py), on the quadrotor tracking problem.  The KKT
    residual must fall below 1e-8 in a handful of iterations, the factorisation must report the right inertia at the
    solution, and the point must be the one scipy finds through the same device model."""
    import os, sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from newton_kkt_demo import newton
    from infiniteexamodels.jl_amd import transcribe, workloads
    from infiniteexamodels.jl_amd.model import ExaModel
    gm = ExaModel(transcribe.exa_core(workloads.quadrotor(400)), device=0)
    x, y, hist = newton(gm, iters=25)
    assert hist[-1]["kkt_residual"] <= 1e-8, hist
    assert len(hist) <= 20
    pos, neg, doubtful = hist[-2]["inertia"]
    assert (pos, neg, doubtful) == (gm.meta.nvar, gm.meta.ncon, 0)      # a minimiser: the reduced Hessian is positive definite
    # feasibility and stationarity re-checked through the oracle on the host
    om = OracleModel(gm.core.to_blob())
    xh, yh = x.cpu().numpy(), y.cpu().numpy()
    assert np.abs(om.cons(xh) - om.lcon).max() <= 1e-8
    assert np.abs(om.grad(xh) + om.jtprod(xh, yh)).max() <= 1e-7
    gm.close()
