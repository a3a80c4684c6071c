"""Run-to-run bitwise reproducibility on the GPU.

obj sums one partial per workgroup in a fixed order; gradient / J'v / Hv entries that MANY items
share (first-stage variables of the farmer and of the stochastic OPF) are reduced per workgroup and
summed by the last workgroup in a fixed order (iem_device.h: iem_shared_*) — no floating-point
atomics whose arrival order could change the rounding; sums over a non-lane axis (pandemic: the u(t) column of
J'v and Hv gets one addend per scenario) are parked row by row and summed in row order by a follow-up kernel
(iem_axis_sum_kernel); whatever would still be a float atomic (collocation stencils, entries reached from two
support grids) is parked per item and summed per entry in the order of a host-built plan (iem_gather_sum_kernel).
Ten calls must give identical bits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _model(name):
    from infiniteexamodels.jl_amd import transcribe, workloads
    mk = {"farmer_40000": lambda: workloads.farmer(40_000), "farmer_900": lambda: workloads.farmer(900),
          "opf_40000": lambda: workloads.opf(40_000), "opf_700": lambda: workloads.opf(700),
          "quadrotor_50000": lambda: workloads.quadrotor(50_000), "pandemic_600x40": lambda: workloads.pandemic(590, 40),
          "quadrotor_oc3_30000": lambda: workloads.quadrotor(30_000, collocation=3), "kinetic_20000": lambda: workloads.kinetic_control(20_000)}[name]
    return transcribe.exa_core(mk())


@pytest.mark.parametrize("name", ["farmer_40000", "farmer_900", "opf_40000", "opf_700", "quadrotor_50000", "pandemic_600x40", "quadrotor_oc3_30000", "kinetic_20000"])
def test_ten_calls_identical_bits(name, built):
    import torch
    from infiniteexamodels.jl_amd.model import ExaModel
    from pyoracle import OracleModel
    core = _model(name)
    blob = core.to_blob()
    gm = ExaModel(core, device=0, blob=blob)
    om = OracleModel(blob)
    om.set_threads(min(om.max_threads(), 16))
    x = om.x0 + 0.1 * np.random.default_rng(0).standard_normal(om.nvar)
    if name.startswith("kinetic"):
        x = om.x0.copy()
    if name.startswith("farmer") or name.startswith("pandemic"):
        x = np.abs(x) + 0.05
    y = np.random.default_rng(1).standard_normal(om.ncon)
    v, vc = np.random.default_rng(2).standard_normal(om.nvar), np.random.default_rng(3).standard_normal(om.ncon)
    xd, yd, vd, vcd = (torch.tensor(a, device="cuda") for a in (x, y, v, vc))
    first = None
    for it in range(10):
        out = (np.float64(gm.obj(xd)).tobytes(), gm.grad(xd).cpu().numpy().tobytes(),
               gm.jtprod(xd, vcd).cpu().numpy().tobytes(), gm.hprod(xd, yd, vd, obj_weight=0.7).cpu().numpy().tobytes())
        if first is None:
            first = out
        else:
            for a, b, what in zip(first, out, ("obj", "grad", "jtprod", "hprod")):
                assert a == b, f"{what} changed between call 0 and call {it}"
    # and they are the right numbers
    g = np.frombuffer(first[1]); jt = np.frombuffer(first[2]); hp = np.frombuffer(first[3])
    np.testing.assert_allclose(g, om.grad(x), rtol=1e-10, atol=1e-10 * max(1.0, np.abs(om.grad(x)).max()))
    np.testing.assert_allclose(jt, om.jtprod(x, vc), rtol=1e-10, atol=1e-10 * max(1.0, np.abs(om.jtprod(x, vc)).max()))
    np.testing.assert_allclose(hp, om.hprod(x, y, v, 0.7), rtol=1e-10, atol=1e-10 * max(1.0, np.abs(om.hprod(x, y, v, 0.7)).max()))
    gm.close()
