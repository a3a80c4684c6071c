"""ExaTranscriptionBackend plug point (src/infiniteopt_backend.jl:85-157, 511-615)."""
import numpy as np
import pytest

import cases
from infiniteexamodels.jl_amd import ExaTranscriptionBackend, InfiniteModel, MI355XBackend


def test_backend_slot_and_rebuild_without_gpu():
    b = ExaTranscriptionBackend(backend=None, print_level=0)
    m, (P1, P2) = cases.rosenbrock()
    m.set_transformation_backend(b)
    b.build_transformation_backend()
    assert b.core is not None and b.model is None and b.options == {"print_level": 0}
    # set_parameter_value(p1, 90.0) → θ updated in place, backend stays ready (test/solve.jl:148-156)
    assert b.update_parameter_value(P1, 90.0) and b.core.theta[b.data.param_mappings[P1].offset] == 90.0
    # start-value updates hit core.x0 (test/solve.jl:211-225)
    m2 = InfiniteModel()
    t = m2.infinite_parameter("t", 0, 1, num_supports=3)
    x = m2.variable("x", t)
    z = m2.variable("z", start=3)
    m2.constraint(x + z == 1)
    b2 = ExaTranscriptionBackend()
    m2.set_transformation_backend(b2)
    b2.build_transformation_backend()
    assert b2.update_start_value(z, 10) and b2.core.x0[int(b2.transformation_variable(z).i) - 1] == 10
    var = b2.transformation_variable(x)
    assert b2.update_start_value(x, 20) and (b2.core.x0[var.offset:var.offset + var.length] == 20).all()
    assert b2.update_start_value(x, lambda t: 42 + 0 * t) and (b2.core.x0[var.offset:var.offset + var.length] == 42).all()
    b2.empty()
    assert b2.core is None and b2.options == {}
    with pytest.warns(UserWarning, match="No previous solution values found"):
        b2.warmstart_backend_start_values()


def test_unknown_backend_is_rejected():
    b = ExaTranscriptionBackend(backend="CUDABackend()")
    m = cases.ode_5x5()
    m.set_transformation_backend(b)
    with pytest.raises(TypeError, match="MI355XBackend"):
        b.build_transformation_backend()


@pytest.mark.gpu
def test_backend_builds_device_model_and_updates_theta(built):
    import torch
    from pyoracle import OracleModel
    b = ExaTranscriptionBackend(backend=MI355XBackend(0))
    m, (pf1, pf2) = cases.pfun()
    m.set_transformation_backend(b)
    b.build_transformation_backend()
    assert b.model is not None and b.model.meta.nvar == 12
    x = torch.tensor(np.linspace(0.5, 1.5, 12), device="cuda")
    f0 = b.model.obj(x)
    assert b.update_parameter_value(pf1, np.cos)          # set_parameter_value(pf1, cos) (test/solve.jl:196)
    np.testing.assert_array_equal(b.model.theta[:3], np.cos([0.0, 0.5, 1.0]))
    om = OracleModel(b.core.to_blob())
    f1 = b.model.obj(x)
    assert f1 != f0 and abs(f1 - om.obj(x.cpu().numpy())) <= 1e-12 * abs(f1)


def test_sharded_backend_refuses_the_solver_slot():
    """A backend holding ONE rank's shard is a build-and-evaluate plug point: optimize() / warm start raise instead of
    handing the global x0 to the rank's local model (no GPU needed: the refusal comes first)."""
    from infiniteexamodels.jl_amd.backend import ExaTranscriptionBackend
    from infiniteexamodels.jl_amd.model import MI355XBackend
    be = ExaTranscriptionBackend(solver=lambda *a, **k: None, backend=MI355XBackend(0, shard=(1, 0, 4)))
    with pytest.raises(NotImplementedError, match="rank 0 of 4"):
        be.optimize()
    with pytest.raises(NotImplementedError, match="build-and-evaluate"):
        be.warmstart_backend_start_values()
