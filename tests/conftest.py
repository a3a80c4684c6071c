import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the host library and the oracle are compiled (both build on CPU)."""
    from infiniteexamodels.jl_amd import lib as iemlib
    import pyoracle
    iemlib.build_library()
    pyoracle.build()
    return True


@pytest.fixture
def lane_fused(built):
    """Force the lane-fused kernels (the path the headline sizes take) on small models: by default
    a support grid of <= 64 workgroups runs its templates side by side (`split_small`)."""
    from infiniteexamodels.jl_amd import lib as iemlib
    with iemlib.options(split_small=0):
        yield


@pytest.fixture(params=["lane_fused", "templates_side_by_side"])
def grid_mode(request, built):
    """Both code shapes of a small model: lane-fused (what large grids use) and the default for
    small grids (one body per template, all in one launch)."""
    from infiniteexamodels.jl_amd import lib as iemlib
    with iemlib.options(split_small=0 if request.param == "lane_fused" else 64):
        yield request.param
