"""COO → CSR assembly: plan on CPU tensors vs scipy; values on the GPU."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import cases
from infiniteexamodels.jl_amd.csr import build_plan
from pyoracle import OracleModel


@pytest.mark.parametrize("name", ["quadrotor_100", "pandemic_20x3", "opf_7", "test_problem_1"])
def test_plan_matches_scipy(name, built):
    om = OracleModel(cases.build_core(name).to_blob())
    x, y = cases.eval_point_for(name, om)
    for which in ("jac", "hess"):
        r, c = om.jac_structure() if which == "jac" else om.hess_structure()
        v = om.jac_coord(x) if which == "jac" else om.hess_coord(x, y, 0.8)
        shape = (om.ncon, om.nvar) if which == "jac" else (om.nvar, om.nvar)
        perm, seg, rowptr, colind = build_plan(torch.from_numpy(r), torch.from_numpy(c), *shape)
        ref = sp.coo_matrix((v, (r, c)), shape=shape).tocsr()
        ref.sum_duplicates()
        ref.sort_indices()
        # the plan keeps structural zeros that scipy also keeps (explicit entries)
        assert np.array_equal(rowptr.numpy(), ref.indptr) and np.array_equal(colind.numpy(), ref.indices)
        vals = np.array([v[perm.numpy()[seg[i]:seg[i + 1]]].sum() for i in range(len(colind))])
        np.testing.assert_allclose(vals, ref.data, rtol=1e-14, atol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["quadrotor_1000", "pandemic_300x7", "opf_600"])
def test_csr_values_on_gpu(name, built):
    from infiniteexamodels.jl_amd.csr import CsrAssembler
    from infiniteexamodels.jl_amd.model import ExaModel
    core = cases.build_core(name)
    gm = ExaModel(core, device=0)
    om = OracleModel(core.to_blob())
    x, y = cases.eval_point_for(name, om)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    for which in ("jac", "hess"):
        asm = CsrAssembler(gm, which)
        coo = gm.jac_coord(xd) if which == "jac" else gm.hess_coord(xd, yd, obj_weight=0.8)
        vals = asm.values(coo).cpu().numpy()
        r, c = om.jac_structure() if which == "jac" else om.hess_structure()
        v = om.jac_coord(x) if which == "jac" else om.hess_coord(x, y, 0.8)
        ref = sp.coo_matrix((v, (r, c)), shape=asm.shape).tocsr()
        ref.sum_duplicates()
        ref.sort_indices()
        assert np.array_equal(asm.rowptr.cpu().numpy(), ref.indptr)
        assert np.array_equal(asm.colind.cpu().numpy(), ref.indices)
        np.testing.assert_allclose(vals, ref.data, rtol=1e-10, atol=1e-10 * max(1.0, np.abs(ref.data).max()))
        # deterministic: bitwise identical on repetition
        assert np.array_equal(vals, asm.values(coo).cpu().numpy())
    gm.close()
