"""The sharded C-ABI on CPU: `iem_shard_blob` (the device-free face of `iem_create_sharded`) cuts a
rank's shard out of the GLOBAL blob in C++ (csrc/iem_shard.hpp).  Checked two ways:

* against the Python transcriber's shards (shard.py re-transcribes the model statement over the
  rank's support window): the two local models must evaluate bit for bit alike — values, structure,
  bounds, starts — and number their variables alike;
* against the global model: shard results, placed with the maps the C-ABI itself reports
  (`iem_shard_var_map`, `iem_shard_template_info`), reassemble the global cons / jac / hess bit for
  bit, every row and COO slot owned exactly once.
Evaluation by the CPU oracle: this tests the cut, not the kernels."""
import numpy as np
import pytest

from infiniteexamodels.jl_amd import lib as iemlib
from infiniteexamodels.jl_amd import shard, transcribe, workloads
from pyoracle import OracleModel

CASES = [("quadrotor", 37, 3, 1), ("quadrotor", 64, 2, 1), ("quadrotor", 11, 8, 1), ("farmer", 23, 4, 1),
         ("pandemic", (9, 7), 3, 2), ("opf", 13, 8, 1), ("pandemic", (9, 7), 7, 2)]


def _global(name, size):
    mk = {"quadrotor": lambda: workloads.quadrotor(size), "farmer": lambda: workloads.farmer(size),
          "opf": lambda: workloads.opf(size), "pandemic": lambda: workloads.pandemic(*size)}[name] if name != "pandemic" else \
        (lambda: workloads.pandemic(*size))
    data = transcribe.ExaMappingData()
    return transcribe.exa_core(mk(), data), data


def _py_shard(name, size, r, world):
    if name == "quadrotor":
        return shard.quadrotor_shard(size, r, world)[0]
    if name == "farmer":
        return shard.farmer_shard(size, r, world)[0]
    if name == "opf":
        return shard.opf_shard(size, r, world)[0]
    return shard.pandemic_shard(size[0], size[1], r, world)[0]


def _point(om, name, seed=0):
    x = om.x0 + 0.1 * np.random.default_rng(seed).standard_normal(om.nvar)
    if name not in ("quadrotor", "opf"):
        x = np.abs(x) + 0.05
    return x, np.random.default_rng(seed + 1).standard_normal(om.ncon)


def _ordinals(t):
    """global item ordinal of every local item of a shard template (item order)."""
    k0, k1, k2 = (np.arange(n) for n in t["dims"])
    g0, g1, g2 = t["global_dims"]
    o = (t["klo"][0] + k0)[None, None, :] + g0 * ((t["klo"][1] + k1)[None, :, None] + g1 * (t["klo"][2] + k2)[:, None, None])
    return o.reshape(-1)


@pytest.mark.parametrize("name,size,world,group", CASES)
def test_cxx_cut_equals_python_shards_and_reassembles(name, size, world, group, built):
    gcore, gdata = _global(name, size)
    gblob = gcore.to_blob()
    G = OracleModel(gblob)
    xg, yg = _point(G, name)
    ref = dict(c=G.cons(xg), j=G.jac_coord(xg), h=G.hess_coord(xg, yg, 0.7), g=G.grad(xg), f=G.obj(xg))
    jr, jc = G.jac_structure()
    hr, hc = G.hess_structure()
    c = np.full(G.ncon, np.nan); j = np.full(G.nnzj, np.nan); h = np.full(G.nnzh, np.nan)
    seen_c = np.zeros(G.ncon, int); seen_j = np.zeros(G.nnzj, int); seen_h = np.zeros(G.nnzh, int)
    owned = np.zeros(G.nvar, int)
    f = 0.0
    for r in range(world):
        lblob, info, vmap, vflag, tpl = iemlib.shard_blob(gblob, group, r, world)
        L = OracleModel(lblob)
        assert (info["nvar"], info["ncon"], info["nnzj"], info["nnzh"]) == (L.nvar, L.ncon, L.nnzj, L.nnzh)
        assert (info["nvar_global"], info["ncon_global"], info["nnzj_global"], info["nnzh_global"]) == (G.nvar, G.ncon, G.nnzj, G.nnzh)
        # --- equal to the Python transcriber's shard, bit for bit -------------------------------
        pcore = _py_shard(name, size, r, world)
        P = OracleModel(pcore.to_blob())
        assert (P.nvar, P.ncon, P.nnzj, P.nnzh) == (L.nvar, L.ncon, L.nnzj, L.nnzh)
        maps = shard.ShardMaps(pcore, gcore, pcore._shard_spec, pcore._shard_data, gdata)
        assert np.array_equal(maps.var_map, vmap)
        assert np.array_equal(maps.replicated, (vflag & 2) != 0) and np.array_equal(maps.var_owned, (vflag & 1) != 0)
        assert info["n_shared"] == int(maps.replicated.sum())
        for arr in ("x0", "lvar", "uvar", "lcon", "ucon"):
            np.testing.assert_array_equal(getattr(L, arr), getattr(P, arr))
        x, y = xg[vmap], None
        for a, b in ((L.jac_structure(), P.jac_structure()), (L.hess_structure(), P.hess_structure())):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        # --- reassembly through the C-ABI's own maps ------------------------------------------------
        row_map = np.full(L.ncon, -1)
        jpos = np.full(L.nnzj, -1)
        hpos = np.full(L.nnzh, -1)
        for t in tpl:
            k = t["ordinals"]
            assert k.size == t["n_items"]
            if t["kind"] == 1:
                row_map[t["o0"] + np.arange(k.size)] = t["global_o0"] + k
                if t["o1step"]:
                    jpos[t["o1"]:t["o1"] + k.size * t["o1step"]] = (t["global_o1"] + t["o1step"] * k[:, None] + np.arange(t["o1step"])[None, :]).reshape(-1)
            if t["o2step"]:
                hpos[t["o2"]:t["o2"] + k.size * t["o2step"]] = (t["global_o2"] + t["o2step"] * k[:, None] + np.arange(t["o2step"])[None, :]).reshape(-1)
        assert (row_map >= 0).all() and (jpos >= 0).all() and (hpos >= 0).all()
        assert np.array_equal(row_map, maps.row_map)
        y = yg[row_map]
        assert np.array_equal(L.cons(x), P.cons(x)) and np.array_equal(L.jac_coord(x), P.jac_coord(x))
        assert np.array_equal(L.hess_coord(x, y, 0.7), P.hess_coord(x, y, 0.7)) and np.array_equal(L.grad(x), P.grad(x))
        assert L.obj(x) == P.obj(x)
        c[row_map] = L.cons(x); seen_c[row_map] += 1
        j[jpos] = L.jac_coord(x); seen_j[jpos] += 1
        h[hpos] = L.hess_coord(x, y, 0.7); seen_h[hpos] += 1
        f += L.obj(x)
        owned[vmap[(vflag & 1) != 0]] += 1
        lr, lc = L.jac_structure()
        assert np.array_equal(row_map[lr], jr[jpos]) and np.array_equal(vmap[lc], jc[jpos])
        lr, lc = L.hess_structure()
        a, b = vmap[lr], vmap[lc]
        assert np.array_equal(np.maximum(a, b), hr[hpos]) and np.array_equal(np.minimum(a, b), hc[hpos])
    assert (seen_c == 1).all() and (seen_j == 1).all() and (seen_h == 1).all()
    assert (owned == 1).all(), "every global variable is owned by exactly one rank"
    assert np.array_equal(c, ref["c"]) and np.array_equal(j, ref["j"]) and np.array_equal(h, ref["h"])
    assert abs(f - ref["f"]) <= 1e-12 * max(1.0, abs(ref["f"]))


def test_unsupported_inputs_fail_loudly(built):
    import cases
    L = iemlib.lib()
    q = cases.build_core("quadrotor_5").to_blob()
    for bad in ((1, 2, 2), (1, -1, 2), (0, 0, 2), (7, 0, 2), (1, 0, 9)):   # rank/world/group out of range, more ranks than supports
        with pytest.raises(iemlib.IemError):
            iemlib.shard_blob(q, *bad)
    w = np.frombuffer(bytearray(q), dtype=np.int64).copy()
    w[9] = 0                                              # no slab table
    with pytest.raises(iemlib.IemError):
        iemlib.shard_blob(w.tobytes(), 1, 0, 2)


@pytest.mark.parametrize("name,size,group", [("quadrotor", 23, 1), ("pandemic", (8, 5), 2), ("pandemic", (8, 5), 1), ("opf", 9, 1)])
def test_world_one_is_the_identity(name, size, group, built):
    """A one-rank "shard" is the whole model: same numbering, same results, no halo, every variable owned."""
    gcore, _ = _global(name, size)
    gblob = gcore.to_blob()
    lblob, info, vmap, vflag, tpl = iemlib.shard_blob(gblob, group, 0, 1)
    G, L = OracleModel(gblob), OracleModel(lblob)
    assert (L.nvar, L.ncon, L.nnzj, L.nnzh) == (G.nvar, G.ncon, G.nnzj, G.nnzh)
    assert info["halo"] == 0 and info["own_lo"] == 0 and info["own_n"] == info["n_global"]
    assert np.array_equal(vmap, np.arange(G.nvar)) and ((vflag & 1) != 0).all() and not (vflag & 4).any()
    x, y = _point(G, name)
    assert np.array_equal(L.cons(x), G.cons(x)) and np.array_equal(L.jac_coord(x), G.jac_coord(x))
    assert np.array_equal(L.hess_coord(x, y, 0.3), G.hess_coord(x, y, 0.3)) and L.obj(x) == G.obj(x)
    for a, b in ((L.jac_structure(), G.jac_structure()), (L.hess_structure(), G.hess_structure())):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert [t["global_index"] for t in tpl] == list(range(G.n_templates))
    lay = shard.ShardLayout.of_cut((lblob, info, vmap, vflag, tpl))
    assert np.array_equal(lay.row_map, np.arange(G.ncon)) and np.array_equal(lay.jac_pos, np.arange(G.nnzj))
    assert np.array_equal(lay.hess_pos, np.arange(G.nnzh)) and lay.owned.all() and not lay.halo.any()


def test_time_sharded_pandemic_keeps_its_stencil_local(built):
    """Pandemic sharded over TIME (group 1) instead of xi: the 2-D slabs are cut along their FAST axis, the
    backward-difference rows need a halo of one time support per xi row, u(t) is itself sharded, and the initial
    conditions s(0, xi) == s0 (a semi-infinite template over xi only) belong to the rank holding t = 0."""
    gcore, _ = _global("pandemic", (9, 4))
    gblob = gcore.to_blob()
    G = OracleModel(gblob)
    xg, yg = _point(G, "pandemic")
    ref = dict(c=G.cons(xg), j=G.jac_coord(xg), h=G.hess_coord(xg, yg, 0.7))
    c = np.full(G.ncon, np.nan); j = np.full(G.nnzj, np.nan); h = np.full(G.nnzh, np.nan)
    f = 0.0
    for r in range(3):
        lblob, info, vmap, vflag, tpl = iemlib.shard_blob(gblob, 1, r, 3)
        assert info["halo"] == (0 if r == 0 else 1) and info["halo_reach"] == 1
        assert info["halo_doubles"] == 8 * 4 + 1        # the window of EVERY slab over t carries the halo support: s,e,i,r and their derivatives (4 xi rows each) + u
        L = OracleModel(lblob)
        x = xg[vmap]
        row_map = np.full(L.ncon, -1); jpos = np.full(L.nnzj, -1); hpos = np.full(L.nnzh, -1)
        for t in tpl:
            k = t["ordinals"]
            if t["kind"] == 1:
                row_map[t["o0"] + np.arange(k.size)] = t["global_o0"] + k
                if t["o1step"]:
                    jpos[t["o1"]:t["o1"] + k.size * t["o1step"]] = (t["global_o1"] + t["o1step"] * k[:, None] + np.arange(t["o1step"])[None, :]).reshape(-1)
            if t["o2step"]:
                hpos[t["o2"]:t["o2"] + k.size * t["o2step"]] = (t["global_o2"] + t["o2step"] * k[:, None] + np.arange(t["o2step"])[None, :]).reshape(-1)
        c[row_map] = L.cons(x); j[jpos] = L.jac_coord(x); h[hpos] = L.hess_coord(x, yg[row_map], 0.7)
        f += L.obj(x)
    assert np.array_equal(c, ref["c"]) and np.array_equal(j, ref["j"]) and np.array_equal(h, ref["h"])
    assert abs(f - G.obj(xg)) <= 1e-12 * max(1.0, abs(G.obj(xg)))


def test_shard_kernels_are_rank_world_and_size_independent(built):
    """A numeric coincidence between, say, a local slab offset and a global data offset depends on the rank
    and the world size; it must not change the SHAPE of the generated source (index values are merged
    per index space only).  Every shard of a time-sharded quadrotor is one of two code objects — the first
    rank's (point constraints; the unsharded model's own kernels) and every later rank's — whatever the
    world size and the horizon, so the in-tree code-object cache serves an 8-GPU run."""
    g_small = transcribe.exa_core(workloads.quadrotor(4000)).to_blob()
    g_big = transcribe.exa_core(workloads.quadrotor(600_000)).to_blob()    # 75 000 supports per shard of 8: the lane-fused regime
    with iemlib.options(split_small=0):
        first = {iemlib.emit_source(iemlib.shard_blob(g_small, 1, 0, 2)[0])[1], iemlib.emit_source(g_small)[1]}
        later = {iemlib.emit_source(iemlib.shard_blob(g_small, 1, 1, 2)[0])[1]}
    for r, w in ((0, 8), (0, 3)):
        first.add(iemlib.emit_source(iemlib.shard_blob(g_big, 1, r, w)[0])[1])
    for r, w in ((1, 8), (3, 8), (7, 8), (1, 2), (2, 3)):
        later.add(iemlib.emit_source(iemlib.shard_blob(g_big, 1, r, w)[0])[1])
    assert len(first) == 1 and len(later) == 1 and first != later
    # the large-grid kernel shape (a function of the grid size: >= 4000 workgroups; lowered here) is again one code object
    # for every first shard and one for every later shard
    with iemlib.options(big_batch_jac=300, big_batch_hess=300):
        big_first = {iemlib.emit_source(iemlib.shard_blob(g_big, 1, 0, w)[0])[1] for w in (2, 3)}
        big_later = {iemlib.emit_source(iemlib.shard_blob(g_big, 1, r, w)[0])[1] for r, w in ((1, 2), (2, 3))}
    assert len(big_first) == 1 and len(big_later) == 1 and big_first != first and big_later != later


def _restricted(supports=None, n=130):
    """tests/cases.irregular: a domain restriction that is not a contiguous range (transform.jl:448-451) + a
    finite variable + a derivative — over all supports, or over a window of them (the Python shard)."""
    from infiniteexamodels.jl_amd.infinite import DomainRestriction, InfiniteModel
    m = InfiniteModel()
    t = m.infinite_parameter("t", 0, 1, supports=supports) if supports is not None else m.infinite_parameter("t", 0, 1, num_supports=n)
    y = m.variable("y", t, start=1.0)
    w = m.variable("w", start=0.5)
    m.constraint(y ** 2 * w >= 2, restriction=DomainRestriction(lambda s: np.sin(40 * s) >= 0.2, t))
    m.constraint(m.deriv(y, t) == -y * w)
    m.objective("min", m.integral(y ** 2, t) + w ** 2)
    return m


@pytest.mark.parametrize("world", [2, 3, 5])
def test_explicit_item_lists_are_filtered_to_the_owned_supports(world, built):
    """A domain-restricted constraint reaches the library as an explicit item list (index COLUMNS, no box).
    The cut keeps the items whose support the rank owns and re-gathers every column; equal to the Python
    transcriber's shard bit for bit, and reassembling the global model through `ordinals`."""
    from infiniteexamodels.jl_amd.infinite import _round_sig
    gdata = transcribe.ExaMappingData()
    gcore = transcribe.exa_core(_restricted(), gdata)
    gblob = gcore.to_blob()
    G = OracleModel(gblob)
    xg = np.abs(G.x0 + 0.1 * np.random.default_rng(0).standard_normal(G.nvar)) + 0.05
    yg = np.random.default_rng(1).standard_normal(G.ncon)
    ref = dict(c=G.cons(xg), j=G.jac_coord(xg), h=G.hess_coord(xg, yg, 0.7))
    c = np.full(G.ncon, np.nan); j = np.full(G.nnzj, np.nan); h = np.full(G.nnzh, np.nan)
    t_g = _round_sig(np.linspace(0.0, 1.0, 130))
    n_explicit = 0
    for r in range(world):
        lblob, info, vmap, vflag, tpl = iemlib.shard_blob(gblob, 1, r, world)
        L = OracleModel(lblob)
        # the Python transcriber's shard of the same statement
        local, f = shard.window(t_g, r, world, halo=1)
        pm = _restricted(supports=local)
        pm.shard = shard.ShardSpec(group_index=1, rank=r, world=world, **f)
        P = OracleModel(transcribe.exa_core(pm, transcribe.ExaMappingData()).to_blob())
        assert (P.nvar, P.ncon, P.nnzj, P.nnzh) == (L.nvar, L.ncon, L.nnzj, L.nnzh)
        x = xg[vmap]
        row_map = np.full(L.ncon, -1); jpos = np.full(L.nnzj, -1); hpos = np.full(L.nnzh, -1)
        for t in tpl:
            k = t["ordinals"]
            n_explicit += t["items_offset"] >= 0
            if t["kind"] == 1:
                row_map[t["o0"] + np.arange(k.size)] = t["global_o0"] + k
                if t["o1step"]:
                    jpos[t["o1"]:t["o1"] + k.size * t["o1step"]] = (t["global_o1"] + t["o1step"] * k[:, None] + np.arange(t["o1step"])[None, :]).reshape(-1)
            if t["o2step"]:
                hpos[t["o2"]:t["o2"] + k.size * t["o2step"]] = (t["global_o2"] + t["o2step"] * k[:, None] + np.arange(t["o2step"])[None, :]).reshape(-1)
        y = yg[row_map]
        for a, b in ((L.cons(x), P.cons(x)), (L.jac_coord(x), P.jac_coord(x)), (L.hess_coord(x, y, 0.7), P.hess_coord(x, y, 0.7)),
                     (L.grad(x), P.grad(x)), (L.lcon, P.lcon), (L.ucon, P.ucon)):
            assert np.array_equal(a, b)
        for a, b in ((L.jac_structure(), P.jac_structure()), (L.hess_structure(), P.hess_structure())):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        c[row_map] = L.cons(x); j[jpos] = L.jac_coord(x); h[hpos] = L.hess_coord(x, y, 0.7)
    assert n_explicit == world
    assert np.array_equal(c, ref["c"]) and np.array_equal(j, ref["j"]) and np.array_equal(h, ref["h"])


@pytest.mark.parametrize("name,world", [("quadrotor_oc3_40", 2), ("quadrotor_oc3_40", 3), ("quadrotor_oc3_700", 8), ("hovercraft_oc4", 3),
                                        ("kinetic_20", 4), ("test_problem_1_oc3", 2)])
def test_orthogonal_collocation_models_shard_by_whole_elements(name, world, built):
    """Orthogonal collocation (ESCAPE34/quadrotor.jl:13-14 — the reference's own benchmark variant): derivative rows
    live on node x element boxes with indices like 2e + j, constant-over-collocation rows on element pairs.  The
    cut takes whole elements (an item goes to the rank owning the last support it references), the halo covers the
    element's first support.  The Python transcriber cannot shard these; the check is the global model itself:
    shard results placed through the C-ABI's maps reproduce cons / jac / hess bit for bit, every entry owned once."""
    import cases
    gblob = cases.build_core(name).to_blob()
    G = OracleModel(gblob)
    xg, yg = cases.eval_point_for(name, G)
    ref = dict(c=G.cons(xg), j=G.jac_coord(xg), h=G.hess_coord(xg, yg, 0.7), f=G.obj(xg), g=G.grad(xg))
    jr, jc = G.jac_structure()
    c = np.full(G.ncon, np.nan); j = np.full(G.nnzj, np.nan); h = np.full(G.nnzh, np.nan)
    seen = [np.zeros(n, int) for n in (G.ncon, G.nnzj, G.nnzh)]
    owned = np.zeros(G.nvar, int)
    f, g = 0.0, np.zeros(G.nvar)
    reach = None
    for r in range(world):
        cut = iemlib.shard_blob(gblob, 1, r, world)
        lblob, info, vmap, vflag, tpl = cut
        lay = shard.ShardLayout.of_cut(cut)
        reach = info["halo_reach"]
        assert info["halo"] == min(reach, info["own_lo"]) and reach >= 2
        L = OracleModel(lblob)
        x, y = xg[vmap], yg[lay.row_map]
        c[lay.row_map] = L.cons(x); j[lay.jac_pos] = L.jac_coord(x); h[lay.hess_pos] = L.hess_coord(x, y, 0.7)
        for sarr, pos in zip(seen, (lay.row_map, lay.jac_pos, lay.hess_pos)):
            sarr[pos] += 1
        owned[vmap[lay.owned]] += 1
        f += L.obj(x)
        np.add.at(g, vmap, L.grad(x))
        lr, lc = L.jac_structure()
        assert np.array_equal(lay.row_map[lr], jr[lay.jac_pos]) and np.array_equal(vmap[lc], jc[lay.jac_pos])
    assert all((sarr == 1).all() for sarr in seen) and (owned == 1).all()
    assert np.array_equal(c, ref["c"]) and np.array_equal(j, ref["j"]) and np.array_equal(h, ref["h"])
    assert abs(f - ref["f"]) <= 1e-12 * max(1.0, abs(ref["f"]))
    np.testing.assert_allclose(g, ref["g"], rtol=1e-13, atol=1e-13)
