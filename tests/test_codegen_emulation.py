"""Generated kernels, compiled for the host against oracle/host_emulation.h and driven with the
library's own launch plan, must reproduce the oracle — a GPU-free check of the generator
(template fusion, symbolic sweeps, slot layout, guards, index arithmetic, block-store
position arithmetic)."""
import numpy as np
import pytest

import cases
from emu import EmulatedModel
from pyoracle import OracleModel


def _rel(a, b):
    if len(b) == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


@pytest.mark.parametrize("name", list(cases.small_cases()))
def test_emulated_kernels_match_oracle(name, grid_mode):
    core = cases.build_core(name)
    blob = core.to_blob()
    om = OracleModel(blob)
    x, y = cases.eval_point_for(name, om)
    em = EmulatedModel(core, blob)
    assert abs(em.obj(x) - om.obj(x)) <= 1e-12 * max(1.0, abs(om.obj(x)))
    assert _rel(em.cons(x), om.cons(x)) <= 1e-14
    assert _rel(em.grad(x), om.grad(x)) <= 1e-14
    j = em.jac_coord(x, om.nnzj)
    h = em.hess_coord(x, y, 0.7, om.nnzh)
    assert not np.isnan(j).any() and not np.isnan(h).any(), "every output slot must be written"
    assert _rel(j, om.jac_coord(x)) <= 1e-14
    assert _rel(h, om.hess_coord(x, y, 0.7)) <= 1e-14
    if any(k["kind"] == 8 for k in em.kernels):     # the fused jac + hess launch (iem_jac_hess_coord): the same BYTES
        jp, hp = em.jac_hess_coord(x, y, 0.7, om.nnzj, om.nnzh)
        assert np.array_equal(jp, j) and np.array_equal(hp, h)
    else:                                           # only models without Jacobian or without Hessian entries have none
        assert om.nnzj == 0 or om.nnzh == 0
    # one launch per solver phase (iem_eval_trial: obj + cons!; iem_eval_accepted: grad! + jac_coord! + hess_coord!): the
    # member kinds' own bodies behind one dispatcher — the same BYTES as the separate calls
    if om.ncon > 0 and em.n_partials > 0:
        assert em.has("trial")
        f, c = em.eval_trial(x)
        assert f == em.obj(x) and np.array_equal(c, em.cons(x))
    if em.has("accepted"):
        g, ja, ha = em.eval_accepted(x, y, 0.7, om.nnzj, om.nnzh)
        assert np.array_equal(g, em.grad(x)) and np.array_equal(ja, j) and np.array_equal(ha, h)
    else:
        assert (om.nnzj == 0) + (om.nnzh == 0) + (not em.has("grad")) >= 2
    if em.has("point"):          # ... and all five of one point in one launch (iem_eval_all)
        f, c, g, ja, ha = em.eval_all(x, y, 0.7, om.nnzj, om.nnzh)
        assert f == em.obj(x) and np.array_equal(c, em.cons(x)) and np.array_equal(g, em.grad(x)) and np.array_equal(ja, j) and np.array_equal(ha, h)
    else:
        assert om.ncon == 0 or em.n_partials == 0
    # matrix-free products (jprod! / jtprod! / hprod!)
    rng = np.random.default_rng(5)
    v, vc = rng.standard_normal(om.nvar), rng.standard_normal(om.ncon)
    assert _rel(em.jprod(x, v), om.jprod(x, v)) <= 1e-13
    assert _rel(em.jtprod(x, vc), om.jtprod(x, vc)) <= 1e-13
    assert _rel(em.hprod(x, y, v, 0.7), om.hprod(x, y, v, 0.7)) <= 1e-13


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_store_modes_emit_and_agree(mode, grid_mode):
    core = cases.build_core("pandemic_20x3")
    blob = core.to_blob()
    om = OracleModel(blob)
    x, y = cases.eval_point_for("pandemic_20x3", om)
    from infiniteexamodels.jl_amd import lib as iemlib
    try:
        em = EmulatedModel(core, blob, store_mode=mode)
        assert _rel(em.jac_coord(x, om.nnzj), om.jac_coord(x)) <= 1e-14
        assert _rel(em.hess_coord(x, y, 1.0, om.nnzh), om.hess_coord(x, y, 1.0)) <= 1e-14
    finally:
        iemlib.set_option("store_mode", iemlib.DEFAULT_STORE_MODE)


def test_generated_source_is_size_independent(built):
    """Sizes are kernel arguments: one code object serves every grid larger than `split_small`
    workgroups (40 000 … 10^6 supports and beyond), a second one every smaller grid."""
    from infiniteexamodels.jl_amd import lib as iemlib, transcribe, workloads
    key = lambda S: iemlib.emit_source(transcribe.exa_core(workloads.quadrotor(S)).to_blob())[1]
    large = {key(S) for S in (40_000, 100_000)}
    small = {key(S) for S in (100, 1000, 4096)}
    assert len(large) == 1 and len(small) == 1 and large != small
    with iemlib.options(split_small=0):
        assert {key(100)} == large       # the lane-fused source is the same at any size
    # LARGE grids (>= 4000 workgroups, about 2e6 supports) get another kernel shape (48-slot staging batch, XCD-aware tile
    # walk): again one code object for all of its sizes — and what build() compiles for it from a small model with lowered
    # thresholds IS what a model of that size asks for at run time
    with iemlib.options(split_small=0, big_batch_jac=1, big_batch_hess=1):
        huge = {key(2000)}
    with iemlib.options(big_batch_jac=500, big_batch_hess=500):      # (a 2.2e6-support model, without building one)
        assert {key(300_000)} == huge and huge != large
    with iemlib.options(big_batch_jac=0, big_batch_hess=0):
        assert {key(300_000)} == large


@pytest.mark.parametrize("name", ["quadrotor_1000", "pandemic_300x7", "opf_600", "quadrotor_oc3_700"])
@pytest.mark.parametrize("big", [dict(big_batch_jac=1, big_batch_hess=0), dict(big_batch_jac=1, big_batch_hess=1), dict(big_batch_jac=1, big_batch_hess=1, big_xcd=0),
                                 dict(big_batch_jac=1, big_batch_hess=1, big_tile=0), dict(big_batch_jac=1, big_batch_hess=1, big_tile=256)])
def test_large_grid_staging_batch_writes_the_same_values(name, big, lane_fused):
    """The large-grid shape of jac_coord! / hess_coord! (1 024-lane tiles in a program whose other kinds keep 512 — two copies
    of the tile-dependent primitives in namespaces —, 48-slot staging batch, XCD-aware tile walk; for jac_coord! only, for both,
    without the remap, at the model's own tile, at a smaller one) against the oracle, stand-alone kernels and the fused pair
    — it moves barriers and tiles, never values."""
    from infiniteexamodels.jl_amd import lib as iemlib
    core = cases.build_core(name)
    blob = core.to_blob()
    om = OracleModel(blob)
    x, y = cases.eval_point_for(name, om)
    ref = EmulatedModel(core, blob)
    with iemlib.options(**big):
        em = EmulatedModel(core, blob)
        assert em.source != ref.source
        j, h = em.jac_coord(x, om.nnzj), em.hess_coord(x, y, 0.7, om.nnzh)
        assert np.array_equal(j, ref.jac_coord(x, om.nnzj)) and np.array_equal(h, ref.hess_coord(x, y, 0.7, om.nnzh))
        assert _rel(j, om.jac_coord(x)) <= 1e-14 and _rel(h, om.hess_coord(x, y, 0.7)) <= 1e-14
        tiles = {k["kind"]: k["block"] for k in em.kernels}
        assert tiles[1] == (big.get("big_tile", 1024) or tiles[0]) and tiles[0] == ref.kernels[0]["block"]     # jac at the big tile, cons! at the model's
        if tiles[1] == tiles[2]:        # both kinds at one workgroup size: they also share the fused launch
            jp, hp = em.jac_hess_coord(x, y, 0.7, om.nnzj, om.nnzh)
            assert np.array_equal(jp, j) and np.array_equal(hp, h)
        else:
            assert not any(k["kind"] == 8 for k in em.kernels)
        # the other kinds of the same (mixed-size) program are untouched
        assert np.array_equal(em.cons(x), ref.cons(x)) and em.obj(x) == ref.obj(x) and np.array_equal(em.grad(x), ref.grad(x))


def test_cross_template_cse(lane_fused):
    """x7 feeds six templates: its load and its sincos appear once per lane in the fused kernel."""
    from infiniteexamodels.jl_amd import lib as iemlib, transcribe, workloads
    src, _ = iemlib.emit_source(transcribe.exa_core(workloads.quadrotor(100)).to_blob())
    jac = src[src.index("void iem_jac_g0b_body"):src.index("void iem_hess_g0")]
    assert jac.count("sincos(") == 3 and jac.count(" tan(") == 1
    assert jac.count("? X[") == 6   # x7, x8, x9, u1, u2, u3 (u4 and every affine row enter linearly: no loads)
    # the rows whose partials are item data (difference rows h, -1, +1; affine dynamics; point constraints) sit in a body of
    # their own (Options::jac_split): no x load, no arithmetic
    data = src[src.index("void iem_jac_g0a_body"):src.index("void iem_jac_g0b_body")]
    assert "X[" not in data.split("{", 1)[1] and "sincos(" not in data and data.count("iem_stage<3>") == 9 and data.count("iem_stage<2>") == 3


@pytest.mark.parametrize("name", ["quadrotor_100", "pandemic_20x3", "opf_7", "operator_zoo", "irregular", "rosenbrock"])
def test_merged_hessian_layout_is_equivalent(name, lane_fused):
    """Opt-in merged layout: fewer entries, same matrix (dense sums equal), lower triangular."""
    from helpers import coo_to_dense
    from infiniteexamodels.jl_amd import lib as iemlib
    core = cases.build_core(name)
    blob = core.to_blob()
    om = OracleModel(blob)
    x, y = cases.eval_point_for(name, om)
    iemlib.set_option("hess_merge", 1)
    try:
        r, c = iemlib.blob_hess_structure(blob)
        h = EmulatedModel(core, blob).hess_coord(x, y, 0.7, len(r))
    finally:
        iemlib.set_option("hess_merge", 0)
    assert len(r) < om.nnzh and (r >= c).all() and not np.isnan(h).any()
    ro, co = om.hess_structure()
    D = coo_to_dense(r, c, h, (om.nvar, om.nvar))
    Do = coo_to_dense(ro, co, om.hess_coord(x, y, 0.7), (om.nvar, om.nvar))
    np.testing.assert_allclose(D, Do, rtol=1e-13, atol=1e-13 * max(1.0, np.abs(Do).max()))


def test_long_second_grid_dimension_is_folded(lane_fused):
    """A 2-D support grid whose second extent exceeds gridDim.y (65535) is folded over blockIdx.z;
    the overshoot blocks must neither load nor store."""
    from infiniteexamodels.jl_amd import transcribe, workloads
    from infiniteexamodels.jl_amd import lib as iemlib
    core = transcribe.exa_core(workloads.pandemic(54, 66000))     # 64 x 66000 supports
    blob = core.to_blob()
    om = OracleModel(blob)
    x = np.abs(om.x0 + 0.1 * np.random.default_rng(0).standard_normal(om.nvar)) + 0.05
    em = EmulatedModel(core, blob)        # default: lanes along t, xi on blockIdx.y/z
    assert any(k["grid"][2] > 1 for k in em.kernels)
    assert _rel(em.cons(x), om.cons(x)) <= 1e-14
    assert _rel(em.jac_coord(x, om.nnzj), om.jac_coord(x)) <= 1e-14


@pytest.mark.parametrize("nt,nxi", [(54, 300), (300, 7), (490, 23), (1000, 3), (70, 41)])
def test_flat_two_dimensional_grids_store_by_ordinal(nt, nxi, lane_fused):
    """Opt-in `flat2d`: 2-D support grids walked by ONE linear lane index (no partly filled workgroup per
    row); the difference templates (t = 2..Nt, a sub-box) are staged and written by item ordinal.  Row
    lengths below, around and above the tile, so that rows start and end anywhere inside a workgroup."""
    from infiniteexamodels.jl_amd import lib as iemlib, transcribe, workloads
    core = transcribe.exa_core(workloads.pandemic(nt, nxi))
    blob = core.to_blob()
    om = OracleModel(blob)
    x = np.abs(om.x0 + 0.1 * np.random.default_rng(0).standard_normal(om.nvar)) + 0.05
    y = np.random.default_rng(1).standard_normal(om.ncon)
    with iemlib.options(flat2d=1):
        em = EmulatedModel(core, blob)
    assert all(k["grid"][1] == 1 and k["grid"][2] == 1 for k in em.kernels)
    assert "iem_flush_ord<" in em.source
    assert _rel(em.cons(x), om.cons(x)) <= 1e-14
    assert _rel(em.jac_coord(x, om.nnzj), om.jac_coord(x)) <= 1e-14
    assert _rel(em.hess_coord(x, y, 1.0, om.nnzh), om.hess_coord(x, y, 1.0)) <= 1e-14
    v = np.random.default_rng(2).standard_normal(om.nvar)
    assert _rel(em.jprod(x, v), om.jprod(x, v)) <= 1e-13


def test_short_first_dimension_uses_flat_lanes(lane_fused):
    """A box whose first dimension cannot fill a wave (12 x 70000 supports; the 2 x (S-1) box
    of an OrthogonalCollocation(3) derivative) is walked by one linear lane index."""
    from infiniteexamodels.jl_amd import transcribe, workloads
    core = transcribe.exa_core(workloads.pandemic(2, 70000))
    blob = core.to_blob()
    om = OracleModel(blob)
    x = np.abs(om.x0 + 0.1 * np.random.default_rng(0).standard_normal(om.nvar)) + 0.05
    y = np.random.default_rng(1).standard_normal(om.ncon)
    em = EmulatedModel(core, blob)
    big = [k for k in em.kernels if k["name"].startswith("iem_jac")]
    assert all(k["grid"][1] == 1 and k["grid"][2] == 1 for k in big)
    assert _rel(em.cons(x), om.cons(x)) <= 1e-14
    assert _rel(em.jac_coord(x, om.nnzj), om.jac_coord(x)) <= 1e-14
    assert _rel(em.hess_coord(x, y, 1.0, om.nnzh), om.hess_coord(x, y, 1.0)) <= 1e-14


def test_gradient_zero_fill_is_fused_when_nothing_accumulates(lane_fused):
    """grad!: 12 of the quadrotor's 22 slabs get no objective contribution.  Every gradient slot
    stores exclusively, so the kernel zeroes those slabs itself (no memset launch); with
    fuse_zero = 0 the launch plan lists the ranges for the runtime instead.  The emulator starts
    from a NaN-poisoned vector either way."""
    from infiniteexamodels.jl_amd import lib as iemlib
    core = cases.build_core("quadrotor_100")
    om = OracleModel(core.to_blob())
    x, _ = cases.eval_point_for("quadrotor_100", om)
    em = EmulatedModel(core)
    assert "iem_zero_fill(OUT" in em.source and not [z for z in em.zero_ranges if z[0] == 4]
    np.testing.assert_allclose(em.grad(x), om.grad(x), rtol=1e-13, atol=1e-13)
    iemlib.set_option("fuse_zero", 0)
    try:
        em0 = EmulatedModel(core)
    finally:
        iemlib.set_option("fuse_zero", 1)
    assert "iem_zero_fill(OUT" not in em0.source
    assert sum(hi - lo for k, lo, hi in em0.zero_ranges if k == 4) == 12 * 100
    np.testing.assert_allclose(em0.grad(x), om.grad(x), rtol=1e-13, atol=1e-13)
    # a model whose objective accumulates into shared entries (farmer: E over scenarios of first-stage terms) keeps the memset
    emf = EmulatedModel(cases.build_core("pandemic_20x3"))
    omf = OracleModel(emf.blob)
    xf, _ = cases.eval_point_for("pandemic_20x3", omf)
    np.testing.assert_allclose(emf.grad(xf), omf.grad(xf), rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("name", ["quadrotor_100", "hovercraft", "pandemic_300x7", "test_problem_1"])
def test_stencil_neighbours_are_pulled_not_scattered(name, lane_fused):
    """jtprod!: a backward-difference row adds into x[i] and into x[i-1], the neighbour lane's entry
    (/root/reference/src/transform.jl:535-557).  With `pull_scatter` (default) the neighbour computes that addend
    itself through a shifted clone of the template and the entry leaves through one exclusive store: fewer (the
    quadrotor: no) float atomics, same values as the scattering form and as the oracle."""
    import re
    from infiniteexamodels.jl_amd import lib as iemlib
    core = cases.build_core(name)
    blob = core.to_blob()
    om = OracleModel(blob)
    x, y = cases.eval_point_for(name, om)
    vc = np.random.default_rng(5).standard_normal(om.ncon)
    v = np.random.default_rng(6).standard_normal(om.nvar)
    atomics = {}
    for ps in (1, 0):
        with iemlib.options(pull_scatter=ps, det_scatter=0):     # det_scatter = 0: what is not pulled stays a float atomic, and can be counted
            em = EmulatedModel(core, blob)
        seg = "".join(re.findall(r"iem_jtprod_g\d+.*?(?=\nextern|\Z)", em.source, flags=re.S))
        atomics[ps] = seg.count("iem_grad_atomic(")
        assert _rel(em.jtprod(x, vc), om.jtprod(x, vc)) <= 1e-13
        assert _rel(em.grad(x), om.grad(x)) <= 1e-14
        assert _rel(em.hprod(x, y, v, 0.7), om.hprod(x, y, v, 0.7)) <= 1e-13
    assert atomics[1] < atomics[0]
    if name == "quadrotor_100":
        assert atomics[1] == 0


@pytest.mark.parametrize("name", ["opf_7", "quadrotor_100", "pandemic_20x3", "irregular"])
@pytest.mark.parametrize("lazy", [0, 1, 2])
def test_lazy_loads_do_not_change_the_products(name, lazy, lane_fused):
    """Product / scatter kernels with many loads (the OPF: 84-98) emit a load where the value is first used
    instead of at the head of the kernel (`lazy_loads`; register pressure, not arithmetic): same results."""
    from infiniteexamodels.jl_amd import lib as iemlib
    core = cases.build_core(name)
    blob = core.to_blob()
    om = OracleModel(blob)
    x, y = cases.eval_point_for(name, om)
    v, vc = np.random.default_rng(5).standard_normal(om.nvar), np.random.default_rng(6).standard_normal(om.ncon)
    with iemlib.options(lazy_loads=lazy, lazy_min_loads=1):
        em = EmulatedModel(core, blob)
    assert _rel(em.grad(x), om.grad(x)) <= 1e-14
    assert _rel(em.jprod(x, v), om.jprod(x, v)) <= 1e-13
    assert _rel(em.jtprod(x, vc), om.jtprod(x, vc)) <= 1e-13
    assert _rel(em.hprod(x, y, v, 0.7), om.hprod(x, y, v, 0.7)) <= 1e-13


def test_axis_sums_are_parked_and_summed_in_row_order(lane_fused):
    """pandemic: u(t) enters every scenario's path rows, so column u(t) of J'v (and of Hv) is a sum over the NON-lane
    axis xi.  With `det_axis` (default) the kernel parks one addend per (t, xi) and the plan lists an axis sum
    {c, k0, n0, rows, off} that the runtime's follow-up kernel reduces in row order — no float atomics on that
    column; with det_axis = 0 it is an atomic add.  Both agree with the oracle."""
    import re
    from infiniteexamodels.jl_amd import lib as iemlib
    core = cases.build_core("pandemic_300x7")
    blob = core.to_blob()
    om = OracleModel(blob)
    x, y = cases.eval_point_for("pandemic_300x7", om)
    vc = np.random.default_rng(5).standard_normal(om.ncon)
    v = np.random.default_rng(6).standard_normal(om.nvar)
    n_axis = {}
    for da in (1, 0):
        with iemlib.options(det_axis=da):
            em = EmulatedModel(core, blob)
            plan = iemlib.emit_launch_plan(blob)
        n_axis[da] = [tuple(int(w) for w in ln.split()[1:]) for ln in plan.splitlines() if ln.startswith("axis ")]
        assert _rel(em.jtprod(x, vc), om.jtprod(x, vc)) <= 1e-13
        assert _rel(em.hprod(x, y, v, 0.7), om.hprod(x, y, v, 0.7)) <= 1e-13
        assert _rel(em.grad(x), om.grad(x)) <= 1e-14
    assert not n_axis[0] and len(n_axis[1]) == 2
    for kind, c, k0, n0, rows, off in n_axis[1]:
        assert n0 == 310 and rows == 7 and k0 == 1          # 300 + 10 time supports, 7 scenarios


@pytest.mark.parametrize("name", ["quadrotor_oc3_40", "kinetic_20", "irregular", "pandemic_20x3", "test_problem_1_oc3", "hovercraft_oc4"])
def test_no_float_atomics_remain(name, grid_mode):
    """`det_scatter`: whatever is neither an exclusive store, a shared entry nor an axis sum — collocation stencils,
    entries reached from two support grids, gathered indices — is parked per item and summed per entry in the order of
    a host-built plan (launch plan: gather / gdest / gseg / gperm) when an entry can get MORE than two addends
    (default 1: two addends commute, those atomics stay) or always (2: the generated source contains no float atomic
    at all).  grad! / jtprod! / hprod! agree with the oracle in every setting."""
    from infiniteexamodels.jl_amd import lib as iemlib
    core = cases.build_core(name)
    blob = core.to_blob()
    om = OracleModel(blob)
    x, y = cases.eval_point_for(name, om)
    vc = np.random.default_rng(5).standard_normal(om.ncon)
    v = np.random.default_rng(6).standard_normal(om.nvar)
    n_atomics = {}
    for ds in (2, 1, 0):
        with iemlib.options(det_scatter=ds, fold_colloc=0):   # (folded collocation rows leave nothing to gather: next test)
            em = EmulatedModel(core, blob)
            plan = iemlib.emit_launch_plan(blob)
        n_atomics[ds] = em.source.count("iem_grad_atomic(OUT") + em.source.count("iem_grad_wave_uniform(OUT")     # calls, not the definitions
        if ds == 2:
            assert n_atomics[2] == 0 and "\ngather " in plan
        if ds == 0:
            assert "\ngather " not in plan and n_atomics[0] > 0
        assert _rel(em.grad(x), om.grad(x)) <= 1e-13
        assert _rel(em.jtprod(x, vc), om.jtprod(x, vc)) <= 1e-13
        assert _rel(em.hprod(x, y, v, 0.7), om.hprod(x, y, v, 0.7)) <= 1e-13
    assert n_atomics[1] <= n_atomics[0]
    if name in ("quadrotor_oc3_40", "kinetic_20"):       # collocation: boundary-node entries get more than two addends
        assert n_atomics[1] < n_atomics[0]


@pytest.mark.parametrize("name", ["quadrotor_oc3_40", "quadrotor_oc3_700", "kinetic_20", "hovercraft_oc4", "pandemic_oc3_40x3"])
def test_collocation_rows_are_folded_onto_the_support_lanes(name, grid_mode):
    """`fold_colloc` (default 1): the scatter kinds evaluate the node x element boxes of orthogonal-collocation derivative
    rows — and the element lists of constant_over_collocation — on the lanes of the support grid itself; every addend of a
    row (its own node, the other nodes of its element, the element's lower boundary = the last node of the element before)
    is computed by the lane that owns the entry, summed in registers with the grid's own rows and stored ONCE.  Nothing of
    jtprod! is left for float atomics, the gather plan or a zero fill on the uniform-grid models; what remains on the
    hovercraft are its point constraints and way-points (a few items against many: deferred, as without collocation).
    examples/pandemic.jl: node x element x scenario boxes on the 2-D t x xi grid.
    fold_colloc = 2 (the default) puts the full boxes on the support lanes for EVERY kind: same values, same COO positions
    as with 1."""
    from infiniteexamodels.jl_amd import lib as iemlib
    core = cases.build_core(name)
    blob = core.to_blob()
    om = OracleModel(blob)
    x, y = cases.eval_point_for(name, om)
    vc = np.random.default_rng(5).standard_normal(om.ncon)
    v = np.random.default_rng(6).standard_normal(om.nvar)
    count = lambda src: src.count("iem_grad_atomic(OUT") + src.count("iem_grad_wave_uniform(OUT")
    with iemlib.options(det_scatter=0, fold_colloc=0):
        before = count(iemlib.emit_source(blob)[0])
    with iemlib.options(det_scatter=0):
        em0 = EmulatedModel(core, blob)
    assert count(em0.source) < before / (2 if name.startswith("pandemic") else 4)     # (pandemic: the rows that hold u(t) constant and the initial conditions remain — deferred with the default options)
    assert _rel(em0.jtprod(x, vc), om.jtprod(x, vc)) <= 1e-13
    em = EmulatedModel(core, blob)
    plan = iemlib.emit_launch_plan(blob)
    if name not in ("hovercraft_oc4", "pandemic_oc3_40x3"):     # (pandemic: u(t) is written from the t x xi grid AND by the rows that hold it constant over an element, on the t grid)
        assert count(em0.source) == 0 and "\ngather " not in plan and "\nzero 6 " not in plan
    assert " && qe >= " in em.source                   # pinned clones: node K of an element evaluates row J
    assert _rel(em.grad(x), om.grad(x)) <= 1e-13
    assert _rel(em.jtprod(x, vc), om.jtprod(x, vc)) <= 1e-13
    assert _rel(em.hprod(x, y, v, 0.7), om.hprod(x, y, v, 0.7)) <= 1e-13
    with iemlib.options(fold_colloc=1):
        em2 = EmulatedModel(core, blob)
    assert _rel(em2.cons(x), om.cons(x)) <= 1e-13
    assert np.array_equal(em2.jac_coord(x, om.nnzj), em.jac_coord(x, om.nnzj))
    assert np.array_equal(em2.hess_coord(x, y, 0.7, om.nnzh), em.hess_coord(x, y, 0.7, om.nnzh))
    assert _rel(em2.jac_coord(x, om.nnzj), om.jac_coord(x)) <= 1e-13
    assert _rel(em2.jprod(x, v), om.jprod(x, v)) <= 1e-13 and _rel(em2.jtprod(x, vc), om.jtprod(x, vc)) <= 1e-13
    assert abs(em2.obj(x) - om.obj(x)) <= 1e-13 * max(1.0, abs(om.obj(x)))


@pytest.mark.parametrize("order", [4, 6, 8])
@pytest.mark.parametrize("family", ["quadrotor", "pandemic"])
def test_fold_across_collocation_orders(family, order, lane_fused):
    """OrthogonalCollocation(order) has order - 1 rows per element: the fold and its pinned clones for 3, 5 rows (a clone per
    row and node: 9, 25 of them per derivative), and the plan-driven gather beyond `fold_max_n` = 6 rows — on a 1-D grid
    (quadrotor) and on a 2-D one (pandemic: collocation x scenarios).  Every kind against the oracle."""
    from infiniteexamodels.jl_amd import lib as iemlib, transcribe, workloads
    im = workloads.quadrotor(40, collocation=order) if family == "quadrotor" else workloads.pandemic(30, 3, collocation=order)
    core = transcribe.exa_core(im)
    blob = core.to_blob()
    om = OracleModel(blob)
    em = EmulatedModel(core, blob)
    plan = iemlib.emit_launch_plan(blob)
    rng = np.random.default_rng(order)
    x = np.abs(om.x0 + 0.1 * rng.standard_normal(om.nvar)) + 0.05
    y, v, vc = rng.standard_normal(om.ncon), rng.standard_normal(om.nvar), rng.standard_normal(om.ncon)
    assert _rel(em.cons(x), om.cons(x)) <= 1e-13 and _rel(em.jac_coord(x, om.nnzj), om.jac_coord(x)) <= 1e-13
    assert _rel(em.hess_coord(x, y, 0.7, om.nnzh), om.hess_coord(x, y, 0.7)) <= 1e-13
    assert _rel(em.jprod(x, v), om.jprod(x, v)) <= 1e-13 and _rel(em.jtprod(x, vc), om.jtprod(x, vc)) <= 1e-13
    assert _rel(em.grad(x), om.grad(x)) <= 1e-13 and _rel(em.hprod(x, y, v, 0.7), om.hprod(x, y, v, 0.7)) <= 1e-13
    folded = " && qe >= " in em.source          # the guard of a pinned clone
    assert folded == (order - 1 <= 6)
    parked = sum(int(l.split()[3]) for l in plan.splitlines() if l.startswith("gather 6 "))
    if folded and family == "quadrotor":
        assert parked == 0
    if not folded:
        assert parked > 1000          # (the rows of every element through the plan)


def test_a_few_items_against_many_are_deferred(lane_fused):
    """pandemic: the initial conditions s(0, xi) = s0 live on the xi grid (7 items per template), the path rows that write
    the same entries on the t x xi grid (2170) — another workgroup of the same launch.  The big slots keep their exclusive
    coalesced stores; the few items are parked and ADDED after the kernels by the plan-driven gather (negative = ~entry
    in gdest): no float atomic in the source, no memset in the plan, same J'v."""
    from infiniteexamodels.jl_amd import lib as iemlib
    core = cases.build_core("pandemic_300x7")
    blob = core.to_blob()
    om = OracleModel(blob)
    x, y = cases.eval_point_for("pandemic_300x7", om)
    vc = np.random.default_rng(5).standard_normal(om.ncon)
    em = EmulatedModel(core, blob)
    plan = iemlib.emit_launch_plan(blob)
    assert em.source.count("iem_grad_atomic(OUT") == 0
    gdest = [int(v) for ln in plan.splitlines() if ln.startswith("gdest") for v in ln.split()[1:]]
    assert len(gdest) == 4 * 7 * 3 // 3 and all(d < 0 for d in gdest)       # 4 initial conditions x 7 scenarios (J'v only: one plan)
    assert not [ln for ln in plan.splitlines() if ln.startswith("zero 6 ")]    # kind 6 = jtprod: nothing left to memset
    assert _rel(em.jtprod(x, vc), om.jtprod(x, vc)) <= 1e-13
    with iemlib.options(det_scatter=0):
        em0 = EmulatedModel(core, blob)
    assert em0.source.count("iem_grad_atomic(OUT") > 0
    assert _rel(em0.jtprod(x, vc), om.jtprod(x, vc)) <= 1e-13


@pytest.mark.parametrize("name", ["quadrotor_1000", "pandemic_300x7", "opf_600", "quadrotor_oc3_700", "kinetic_20", "irregular"])
def test_a_carried_halo_exchange_leaves_every_tile_in_place(name, grid_mode):
    """iem_halo_exchange_async: the deferred exchange rides on an evaluation launch as ONE EXTRA LEADING workgroup (column
    0 of the grid; one per row on 2-D grids, only the first works).  Launched that way (the exchange itself stubbed: it
    counts its lanes), every kernel kind that can carry must write exactly what it writes alone — cons!, jac_coord!,
    hess_coord!, jprod!, the fused pair and obj (whose tile walkers and partial count must not see the extra workgroup)."""
    core = cases.build_core(name)
    blob = core.to_blob()
    om = OracleModel(blob)
    x, y = cases.eval_point_for(name, om)
    v = np.random.default_rng(5).standard_normal(om.nvar)
    from infiniteexamodels.jl_amd import lib as iemlib
    ref = EmulatedModel(core, blob)
    assert "iem_halo_wg(*A.comm" not in ref.source      # an unsharded handle's kernels have no carrier prologue at all
    with iemlib.options(carrier=1):                     # (what iem_create_sharded sets)
        em = EmulatedModel(core, blob)
    assert "iem_halo_wg(*A.comm" in em.source
    em.carry = True
    block = em.kernels[0]["block"]
    n = 0
    for got, want in ((em.cons(x), ref.cons(x)), (em.jac_coord(x, om.nnzj), ref.jac_coord(x, om.nnzj)),
                      (em.hess_coord(x, y, 0.7, om.nnzh), ref.hess_coord(x, y, 0.7, om.nnzh)), (em.jprod(x, v), ref.jprod(x, v))):
        assert np.array_equal(got, want)
    n += sum(1 for k in em.kernels if k["kind"] in (0, 1, 2, 5))
    assert em.obj(x) == ref.obj(x)
    n += sum(1 for k in em.kernels if k["kind"] == 3)
    if any(k["kind"] == 8 for k in em.kernels):
        jp, hp = em.jac_hess_coord(x, y, 0.7, om.nnzj, om.nnzh)
        assert np.array_equal(jp, ref.jac_coord(x, om.nnzj)) and np.array_equal(hp, ref.hess_coord(x, y, 0.7, om.nnzh))
        n += 1
    # the solver-phase launches carry too (the accepted-point one unless its grad! member reduces shared entries)
    f, c = em.eval_trial(x)
    assert f == ref.obj(x) and np.array_equal(c, ref.cons(x))
    n += 1
    g, ja, ha = em.eval_accepted(x, y, 0.7, om.nnzj, om.nnzh)
    assert np.array_equal(g, ref.grad(x)) and np.array_equal(ja, ref.jac_coord(x, om.nnzj)) and np.array_equal(ha, ref.hess_coord(x, y, 0.7, om.nnzh))
    n += sum(1 for k in em.kernels if k["kind"] == 10 and em._can_carry(k))
    f, c, g, ja, ha = em.eval_all(x, y, 0.7, om.nnzj, om.nnzh)
    assert f == ref.obj(x) and np.array_equal(c, ref.cons(x)) and np.array_equal(g, ref.grad(x)) and np.array_equal(ja, ref.jac_coord(x, om.nnzj)) and np.array_equal(ha, ref.hess_coord(x, y, 0.7, om.nnzh))
    n += sum(1 for k in em.kernels if k["kind"] == 11 and em._can_carry(k))
    assert em.carried == n * block, "exactly one workgroup per carrying launch ran the exchange"
