"""Sharding (multi-GPU path) on CPU: the shards of a model, evaluated independently and
assembled through ShardMaps, reproduce the global model bit for bit; the only collective
is the objective / replicated-gradient sum (run here over gloo with world_size 2)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from infiniteexamodels.jl_amd import shard, transcribe, workloads
from pyoracle import OracleModel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _global(name, size):
    data = transcribe.ExaMappingData()
    if name == "quadrotor":
        core = transcribe.exa_core(workloads.quadrotor(size), data)
    elif name == "farmer":
        core = transcribe.exa_core(workloads.farmer(size), data)
    elif name == "opf":
        core = transcribe.exa_core(workloads.opf(size), data)
    else:
        core = transcribe.exa_core(workloads.pandemic(size[0], size[1]), data)
    return core, data


def _shard(name, size, r, world):
    if name == "quadrotor":
        core, _ = shard.quadrotor_shard(size, r, world)
    elif name == "farmer":
        core, _ = shard.farmer_shard(size, r, world)
    elif name == "opf":
        core, _ = shard.opf_shard(size, r, world)
    else:
        core, _ = shard.pandemic_shard(size[0], size[1], r, world)
    return core


def _point(om, name, seed=0):
    x = om.x0 + 0.1 * np.random.default_rng(seed).standard_normal(om.nvar)
    if name not in ("quadrotor", "opf"):
        x = np.abs(x) + 0.05
    y = np.random.default_rng(seed + 1).standard_normal(om.ncon)
    return x, y


@pytest.mark.parametrize("name,size,world", [("quadrotor", 37, 3), ("quadrotor", 64, 2), ("quadrotor", 11, 8),
                                             ("farmer", 23, 4), ("pandemic", (9, 7), 3), ("opf", 13, 8)])
def test_shards_reassemble_to_global(name, size, world, built):
    gcore, gdata = _global(name, size)
    G = OracleModel(gcore.to_blob())
    xg, yg = _point(G, name)
    ref = dict(f=G.obj(xg), g=G.grad(xg), c=G.cons(xg), j=G.jac_coord(xg), h=G.hess_coord(xg, yg, 0.7))
    jr, jc = G.jac_structure()
    hr, hc = G.hess_structure()
    f = 0.0
    g = np.zeros(G.nvar)
    c = np.full(G.ncon, np.nan)
    j = np.full(G.nnzj, np.nan)
    h = np.full(G.nnzh, np.nan)
    rows_seen = np.zeros(G.ncon, dtype=int)
    for r in range(world):
        core = _shard(name, size, r, world)
        L = OracleModel(core.to_blob())
        maps = shard.ShardMaps(core, gcore, core._shard_spec, core._shard_data, gdata)
        assert np.array_equal(shard.replicated_indices(core), np.nonzero(maps.replicated)[0])   # shard-only derivation
        x = xg[maps.var_map]
        y = yg[maps.row_map]
        f += L.obj(x)
        np.add.at(g, maps.var_map, L.grad(x))
        c[maps.row_map] = L.cons(x)
        rows_seen[maps.row_map] += 1
        jp = maps.jac_positions(L.template_info, G.template_info)
        hp = maps.hess_positions(L.template_info, G.template_info)
        j[jp] = L.jac_coord(x)
        h[hp] = L.hess_coord(x, y, 0.7)
        # structure: local indices map onto the global structure at the mapped positions
        lr, lc = L.jac_structure()
        assert np.array_equal(maps.row_map[lr], jr[jp]) and np.array_equal(maps.var_map[lc], jc[jp])
        lr, lc = L.hess_structure()
        a, b = maps.var_map[lr], maps.var_map[lc]
        assert np.array_equal(np.maximum(a, b), hr[hp]) and np.array_equal(np.minimum(a, b), hc[hp])
    assert (rows_seen == 1).all(), "every constraint row is owned by exactly one rank"
    assert np.array_equal(c, ref["c"])
    assert np.array_equal(j, ref["j"])
    assert np.array_equal(h, ref["h"])
    assert abs(f - ref["f"]) <= 1e-12 * max(1.0, abs(ref["f"]))
    np.testing.assert_allclose(g, ref["g"], rtol=1e-13, atol=1e-13)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("name,size,world", [("farmer", 40, 2), ("quadrotor", 50, 2), ("farmer", 41, 4), ("quadrotor", 53, 3)])
def test_obj_grad_allreduce_gloo_world2(name, size, world, built):
    """world_size 2 (and 3, 4 with ragged shards) over gloo: obj and replicated-gradient entries via
    ONE small all-reduce."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), name, str(size)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert "OK" in outs[0], outs[0]


@pytest.mark.parametrize("name,size,world,mode", [("quadrotor", 50, 2, "export"), ("quadrotor", 53, 3, "connect"), ("farmer", 40, 2, "forced"),
                                                  ("opf", 30, 2, "connect")])
def test_comm_fallback_gloo(name, size, world, mode, built):
    """when a rank cannot export / map a mailbox, every rank falls back to torch.distributed for the
    halo exchange and the one small all-reduce (shard.ShardComm) — world 2 and 3 over gloo, no GPU."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "comm_fallback_worker.py"), name, str(size), mode],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert "OK rccl" in outs[0], outs[0]
