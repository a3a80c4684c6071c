"""KKT assembly + rocSOLVER re-factorisation against scipy on the host."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.sparse.linalg import spsolve

import cases
from pyoracle import OracleModel

pytestmark = pytest.mark.gpu


def host_kkt(om, x, y, sigma, dw, dc, w=1.0):
    n, m = om.nvar, om.ncon
    hr, hc = om.hess_structure()
    jr, jc = om.jac_structure()
    L = sp.coo_matrix((om.hess_coord(x, y, w), (hr, hc)), shape=(n, n)).tocsr()
    H = L + L.T - sp.diags(L.diagonal())
    J = sp.coo_matrix((om.jac_coord(x), (jr, jc)), shape=(m, n)).tocsr()
    return sp.bmat([[H + sp.diags(sigma + dw), J.T], [J, -dc * sp.identity(m)]]).tocsr()


@pytest.mark.parametrize("name", ["quadrotor_100", "pandemic_20x3", "opf_7", "farmer_5"])
def test_kkt_assembly_refactor_solve(name, built):
    import torch
    from infiniteexamodels.jl_amd.kkt import KKTSystem
    from infiniteexamodels.jl_amd.model import ExaModel
    core = cases.build_core(name)
    blob = core.to_blob()
    om = OracleModel(blob)
    gm = ExaModel(core, device=0, blob=blob)
    kkt = KKTSystem(gm)
    rng = np.random.default_rng(3)
    n, m = om.nvar, om.ncon
    dw, dc = 1e-2, 1e-6

    def at(seed):
        x, y = cases.eval_point_for(name, om, seed)
        sigma = 0.5 + rng.random(n)
        xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
        kkt.assemble(gm.hess_coord(xd, yd, obj_weight=1.0), gm.jac_coord(xd), torch.tensor(sigma, device="cuda"), dw, dc)
        return host_kkt(om, x, y, sigma, dw, dc)

    # assembly: device CSR == host matrix
    Kh = at(0)
    Kd = kkt.to_scipy()
    assert abs(Kd - Kh).max() <= 1e-12 * max(1.0, abs(Kh).max())
    # analysis on the host at this point, then re-factorisation on the device at two NEW points
    kkt.analyse()
    for seed in (5, 9):
        Kh = at(seed)
        kkt.factor()
        rhs = rng.standard_normal(n + m)
        sol = kkt.solve(torch.tensor(rhs, device="cuda")).cpu().numpy()
        ref = spsolve(Kh.tocsc(), rhs)
        res = np.abs(Kh @ sol - rhs).max() / max(1.0, np.abs(rhs).max())
        assert res <= 1e-8, (name, seed, res)
        np.testing.assert_allclose(sol, ref, rtol=1e-6, atol=1e-8 * max(1.0, np.abs(ref).max()))
    kkt.close()
    gm.close()
