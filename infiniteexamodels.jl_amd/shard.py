"""Support-axis sharding for multi-GPU runs — one process per GPU.

The reference is single-process, single-device; its SIMD templates, however, are
embarrassingly parallel over items: every item of a template owns its constraint row and
its COO slots, so ``cons!``, ``jac_coord!`` and ``hess_coord!`` of a support block need no
communication, and only the scalar objective and the gradient entries of replicated
(finite / first-stage) variables need a sum across ranks.

A shard is the SAME model statement transcribed over a window of one infinite
parameter's supports:

* rank ``r`` owns the contiguous support block ``[a_r, b_r)`` of the sharded parameter;
  its window additionally holds ``halo`` supports before ``a_r`` (1 for the backward
  finite-difference rows, ``transform.jl:535-557``) so stencil neighbours are local;
* templates that iterate over the sharded parameter are cut to the owned supports
  (:func:`owned`), measure coefficients come from the GLOBAL grid, templates that do not
  involve the sharded parameter (point constraints, first-stage rows, finite objective
  terms) live on rank 0 only; finite variables are replicated;
* :class:`ShardMaps` gives the local→global maps of variables, constraint rows and COO
  positions used by the tests (and by a consumer that wants the assembled arrays).

Host-side collectives (``torch.distributed``: RCCL on GPUs, gloo in the CPU tests) are
confined to :func:`allreduce_obj_grad`.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

from . import transcribe
from .core import ExaCore, T_CON, T_OBJ
from .infinite import InfiniteModel, _round_sig


@dataclass
class ShardSpec:
    """Attached to an :class:`InfiniteModel` as ``model.shard``; read by :mod:`transcribe`."""
    group_index: int            # 1-based index of the sharded parameter group
    rank: int
    world: int
    own_lo: int                 # local index of the first owned support (= halo actually present)
    own_n: int                  # number of owned supports
    global_lo: int              # global index of the first LOCAL support (window start)
    n_global: int
    coeffs: Optional[np.ndarray] = None   # measure coefficients of the LOCAL supports from the global grid


def partition(n: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous near-equal blocks ``[a_r, b_r)``."""
    base, rem = divmod(n, world)
    out, a = [], 0
    for r in range(world):
        b = a + base + (1 if r < rem else 0)
        out.append((a, b))
        a = b
    return out


def trapezoid_weights(s: np.ndarray) -> np.ndarray:
    d = np.diff(s)
    c = np.zeros_like(s)
    c[:-1] += d / 2
    c[1:] += d / 2
    return c


def window(supports: np.ndarray, rank: int, world: int, halo: int, measure: str = "trapezoid"):
    """Local support window of ``rank`` and its :class:`ShardSpec` fields."""
    n = len(supports)
    a, b = partition(n, world)[rank]
    h = min(halo, a)
    if measure == "trapezoid":
        cg = trapezoid_weights(supports)
    else:
        cg = np.full(n, 1.0 / n)
    return supports[a - h:b], dict(own_lo=h, own_n=b - a, global_lo=a - h, n_global=n, coeffs=cg[a - h:b].copy())


def quadrotor_shard(S_global: int, rank: int, world: int, backend=None):
    """Time-sharded quadrotor (``examples/quadrotor.jl``): returns ``(core, owned supports)``."""
    from . import workloads
    t_g = _round_sig(np.linspace(0.0, 60.0, S_global))
    local, f = window(t_g, rank, world, halo=1)
    im = workloads.quadrotor(supports=local, backend=backend)
    im.shard = ShardSpec(group_index=1, rank=rank, world=world, **f)
    data = transcribe.ExaMappingData()
    core = transcribe.exa_core(im, data)
    core._shard_data = data
    core._shard_spec = im.shard
    return core, f["own_n"]


def farmer_shard(num_scenarios: int, rank: int, world: int, seed: int = 42, backend=None):
    """Scenario-sharded two-stage farmer (``examples/2stage_example.jl``)."""
    from . import workloads
    supp = workloads.farmer_supports(num_scenarios, seed)
    a, b = partition(num_scenarios, world)[rank]
    im = workloads.farmer(supports=supp[a:b], backend=backend)
    im.shard = ShardSpec(group_index=1, rank=rank, world=world, own_lo=0, own_n=b - a, global_lo=a,
                         n_global=num_scenarios, coeffs=np.full(b - a, 1.0 / num_scenarios))
    data = transcribe.ExaMappingData()
    core = transcribe.exa_core(im, data)
    core._shard_data = data
    core._shard_spec = im.shard
    return core, b - a


def opf_shard(num_scenarios: int, rank: int, world: int, seed: int = 0, backend=None):
    """Scenario-sharded stochastic AC-OPF (``ESCAPE34/opf.jl``): the first-stage network lives on
    rank 0, first-stage variables are replicated, ramping rows couple them to every scenario."""
    from . import workloads
    supp = workloads.opf_supports(num_scenarios, seed)
    a, b = partition(num_scenarios, world)[rank]
    im = workloads.opf(supports=supp[a:b], backend=backend)
    im.shard = ShardSpec(group_index=1, rank=rank, world=world, own_lo=0, own_n=b - a, global_lo=a,
                         n_global=num_scenarios, coeffs=None)
    data = transcribe.ExaMappingData()
    core = transcribe.exa_core(im, data)
    core._shard_data = data
    core._shard_spec = im.shard
    return core, b - a


def pandemic_shard(num_supports: int, num_scenarios: int, rank: int, world: int, backend=None):
    """ξ-sharded pandemic SIR (``ESCAPE34/pandemic.jl``): t-stencils stay local, ``u(t)`` and
    the objective ∫u dt are replicated / rank-0 only."""
    from . import workloads
    xi_g = _round_sig(np.linspace(0.1, 0.6, num_scenarios))
    a, b = partition(num_scenarios, world)[rank]
    im = workloads.pandemic(num_supports, xi_supports=xi_g[a:b], backend=backend)
    im.shard = ShardSpec(group_index=2, rank=rank, world=world, own_lo=0, own_n=b - a, global_lo=a,
                         n_global=num_scenarios, coeffs=None)
    data = transcribe.ExaMappingData()
    core = transcribe.exa_core(im, data)
    core._shard_data = data
    core._shard_spec = im.shard
    return core, b - a


# ---------------------------------------------------------------------------
# local → global maps
# ---------------------------------------------------------------------------
class ShardMaps:
    """Index maps between a shard's ExaCore and the global model's ExaCore.

    Both cores must come from the same model statement (same template tags)."""

    def __init__(self, local: ExaCore, glob: ExaCore, spec: ShardSpec, local_data, global_data):
        self.spec = spec
        self._glob_templates = list(glob.templates)
        g = spec.group_index
        # variables -----------------------------------------------------------------
        self.var_map = np.full(local.nvar, -1, dtype=np.int64)   # 0-based local -> 0-based global
        self.var_owned = np.zeros(local.nvar, dtype=bool)
        self.replicated = np.zeros(local.nvar, dtype=bool)   # finite / non-sharded variables held by every rank
        lv = list(local_data.finvar_slabs) + list(local_data.infvar_slabs)
        gv = list(global_data.finvar_slabs) + list(global_data.infvar_slabs)
        assert len(lv) == len(gv)
        for (lvar, lgroups), (gvar, ggroups) in zip(lv, gv):
            assert lgroups == ggroups
            if g not in lgroups:
                idx = np.arange(lvar.length)
                self.var_map[lvar.offset + idx] = gvar.offset + idx
                self.var_owned[lvar.offset + idx] = spec.rank == 0
                self.replicated[lvar.offset + idx] = True
                continue
            ax = lgroups.index(g)
            lshape, gshape = lvar.size, gvar.size
            lidx = np.indices(lshape).reshape(len(lshape), -1)
            gidx = lidx.copy()
            gidx[ax] += spec.global_lo
            lflat = np.ravel_multi_index(tuple(lidx), lshape, order="F")
            gflat = np.ravel_multi_index(tuple(gidx), gshape, order="F")
            self.var_map[lvar.offset + lflat] = gvar.offset + gflat
            self.var_owned[lvar.offset + lflat] = lidx[ax] >= spec.own_lo
        assert (self.var_map >= 0).all()
        # templates -------------------------------------------------------------------
        gt = {t.tag: t for t in glob.templates}
        self.row_map = np.full(local.ncon, -1, dtype=np.int64)
        self.pairs = []   # (local template, global template, global item ordinal per local item)
        for t in local.templates:
            G = gt[t.tag]
            kmap = self._item_map(t, G, spec)
            self.pairs.append((t, G, kmap))
            if t.kind == T_CON:
                self.row_map[t.o0 + np.arange(len(t.items))] = G.o0 + kmap
        assert (self.row_map >= 0).all()

    @staticmethod
    def _item_map(t, G, spec) -> np.ndarray:
        """global item ordinal of each local item, via the templates' group-index columns."""
        alias = f"group_idx{spec.group_index}"
        n = len(t.items)
        if alias not in t.items.fields:
            return np.arange(n, dtype=np.int64)
        # every item is identified by its tuple of integer fields; shift the sharded one
        names = [k for k, f in t.items.fields.items() if f.kind == "int"]
        key_l = np.stack([t.items.column(k) + (spec.global_lo if k == alias else 0) for k in names], axis=1)
        key_g = np.stack([G.items.column(k) for k in names], axis=1)
        lookup = {tuple(r): i for i, r in enumerate(key_g.tolist())}
        return np.array([lookup[tuple(r)] for r in key_l.tolist()], dtype=np.int64)

    def jac_positions(self, local_info, global_info):
        """global COO position of every local Jacobian entry.
        ``*_info(i)`` → dict with o1/o1step for template i (``ExaModel.template_info``)."""
        return self._positions(local_info, global_info, "o1", "o1step", cons_only=True)

    def hess_positions(self, local_info, global_info):
        return self._positions(local_info, global_info, "o2", "o2step", cons_only=False)

    def _positions(self, linfo, ginfo, okey, skey, cons_only):
        out = []
        gl_index = {id(G): i for i, G in enumerate(self._glob_templates)}
        for li, (t, G, kmap) in enumerate(self.pairs):
            if cons_only and t.kind != T_CON:
                continue
            a, b = linfo(li), ginfo(gl_index[id(G)])
            step = a[skey]
            assert step == b[skey]
            if step == 0:
                continue
            pos = b[okey] + step * kmap[:, None] + np.arange(step)[None, :]
            out.append((a[okey], pos.reshape(-1)))
        res = np.concatenate([p for _, p in sorted(out, key=lambda z: z[0])]) if out else np.zeros(0, np.int64)
        return res

    def bind(self, glob: ExaCore):
        self._glob_templates = list(glob.templates)
        return self


def replicated_indices(core: ExaCore) -> np.ndarray:
    """0-based local indices of the variables every rank holds a copy of (finite variables and
    infinite variables that do not depend on the sharded parameter) — the gradient entries that
    need the all-reduce.  Computed from the shard alone (no global core)."""
    g = core._shard_spec.group_index
    d = core._shard_data
    out = [np.arange(v.offset, v.offset + v.length) for v, groups in list(d.finvar_slabs) + list(d.infvar_slabs)
           if g not in groups]
    return np.concatenate(out).astype(np.int64) if out else np.zeros(0, dtype=np.int64)


class ShardLayout:
    """Local → global maps of a shard as the C-ABI reports them (``iem_shard_var_map``,
    ``iem_shard_template_info`` / ``_items``): ``var_map``, the ``owned / replicated / halo`` masks,
    ``row_map`` (constraint rows) and the global COO positions ``jac_pos`` / ``hess_pos`` of every local
    Jacobian / Hessian entry.  Built from a sharded :class:`ExaModel` or from the tuple ``lib.shard_blob`` returns."""

    def __init__(self, var_map, var_flag, templates, ncon: int, nnzj: int, nnzh: int):
        self.var_map = np.asarray(var_map)
        f = np.asarray(var_flag)
        self.owned, self.replicated, self.halo = (f & 1) != 0, (f & 2) != 0, (f & 4) != 0
        self.row_map = np.full(ncon, -1, dtype=np.int64)
        self.jac_pos = np.full(nnzj, -1, dtype=np.int64)
        self.hess_pos = np.full(nnzh, -1, dtype=np.int64)
        for t in templates:
            k = t["ordinals"]
            if t["kind"] == T_CON:
                self.row_map[t["o0"] + np.arange(k.size)] = t["global_o0"] + k
                if t["o1step"]:
                    self.jac_pos[t["o1"]:t["o1"] + k.size * t["o1step"]] = \
                        (t["global_o1"] + t["o1step"] * k[:, None] + np.arange(t["o1step"])[None, :]).reshape(-1)
            if t["o2step"]:
                self.hess_pos[t["o2"]:t["o2"] + k.size * t["o2step"]] = \
                    (t["global_o2"] + t["o2step"] * k[:, None] + np.arange(t["o2step"])[None, :]).reshape(-1)
        assert (self.row_map >= 0).all() and (self.jac_pos >= 0).all() and (self.hess_pos >= 0).all()

    @classmethod
    def of_model(cls, gm) -> "ShardLayout":
        vm, vf = gm.shard_var_map()
        return cls(vm, vf, gm.shard_templates(), gm.meta.ncon, gm.meta.nnzj, gm.meta.nnzh)

    @classmethod
    def of_cut(cls, cut) -> "ShardLayout":
        _, info, vm, vf, tpl = cut
        return cls(vm, vf, tpl, info["ncon"], info["nnzj"], info["nnzh"])


def connect_mailboxes(gm, dist=None) -> None:
    """Wire the ranks' mailboxes (``iem_comm_export`` → all-gather of the 128-byte handles over the
    host process group, like an ncclUniqueId → ``iem_comm_connect``); afterwards
    ``gm.halo_exchange`` / ``gm.allreduce_obj_grad`` run without torch.distributed."""
    if dist is None:
        import torch.distributed as dist
    mine = gm.comm_export()
    handles = [None] * dist.get_world_size()
    dist.all_gather_object(handles, mine)
    gm.comm_connect(b"".join(handles))
    dist.barrier()


class ShardComm:
    """The two exchanges a sharded handle needs around its evaluation calls, with the fallback ``north_star`` names:

    * ``kind == "own"`` — the library's mailbox kernels (``iem_halo_exchange[_async]``, ``iem_allreduce_obj_grad``:
      one-shot pushes over HIP IPC / xGMI), when EVERY rank could export and map them;
    * ``kind == "rccl"`` — ``torch.distributed`` (backend ``nccl`` = RCCL on the GPUs, ``gloo`` in the CPU tests): the
      8·(1 + n_shared)-byte all-reduce of :func:`allreduce_obj_grad_device`, and a point-to-point send / receive of the
      halo doubles.  Taken when ``iem_comm_export`` / ``iem_comm_connect`` fails on any rank (no fine-grained IPC memory
      between different GPUs: ``mailbox_kind == 2``), or when ``force_fallback`` is set.

    All ranks agree on the kind (a gathered flag) before anything is exchanged.  ``why`` says what forced the fallback.
    The fallback only needs ``gm.shard_var_map()`` / ``gm.shard_info()`` and tensors of ``x``'s device, so it runs on CPU
    tensors under gloo (``tests/test_shard.py``)."""

    def __init__(self, gm, dist=None, force_fallback: bool = False):
        import torch
        if dist is None:
            import torch.distributed as dist
        self.gm, self.dist, self._torch = gm, dist, torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        err = "forced" if force_fallback else ""
        mine = b""
        if not force_fallback:
            try:
                mine = gm.comm_export()
            except Exception as e:     # noqa: BLE001 — becomes the reason of the fallback
                err = f"export: {e}"
        handles = [None] * self.world
        dist.all_gather_object(handles, (mine, err))
        errs = [h[1] for h in handles if h[1]]
        if not errs:
            try:
                gm.comm_connect(b"".join(h[0] for h in handles))
            except Exception as e:     # noqa: BLE001
                err = f"connect: {e}"
            flags = [None] * self.world
            dist.all_gather_object(flags, err)
            errs = [f for f in flags if f]
        self.kind = "rccl" if errs else "own"
        self.why = errs[0] if errs else ""
        if self.kind == "own":
            return
        # fallback plan: who sends what to whom (halo copies are the left neighbour's LAST owned supports)
        vm, vf = gm.shard_var_map()
        info = gm.shard_info()
        halo = (vf & 4) != 0
        self._dst = np.nonzero(halo)[0].astype(np.int64)
        wanted = [None] * self.world
        dist.all_gather_object(wanted, vm[halo].astype(np.int64))
        self._left = self.rank - 1 if self.rank > 0 and self._dst.size else None
        self._right, self._src = None, np.zeros(0, np.int64)
        if self.rank + 1 < self.world and wanted[self.rank + 1].size:
            order = np.argsort(vm, kind="stable")
            pos = np.searchsorted(vm[order], wanted[self.rank + 1])
            src = order[np.minimum(pos, vm.size - 1)]
            if not (np.array_equal(vm[src], wanted[self.rank + 1]) and ((vf[src] & 1) != 0).all()):
                raise RuntimeError("ShardComm: the right neighbour's halo copies are not all owned by this rank")
            self._right, self._src = self.rank + 1, src.astype(np.int64)
        self._shared = np.nonzero((vf & 2) != 0)[0].astype(np.int64)
        self._n_shared = int(info["n_shared"])
        self._bufs = {}

    def _dev(self, ref):
        """Index / staging tensors on the device of ``ref`` (built once per device)."""
        key = str(ref.device)
        if key not in self._bufs:
            t = self._torch
            mk = lambda a: t.as_tensor(a, device=ref.device)
            self._bufs[key] = dict(src=mk(self._src), dst=mk(self._dst), shared=mk(self._shared),
                                   send=t.empty(self._src.size, dtype=t.float64, device=ref.device),
                                   recv=t.empty(self._dst.size, dtype=t.float64, device=ref.device),
                                   red=t.empty(1 + self._shared.size, dtype=t.float64, device=ref.device))
        return self._bufs[key]

    def halo_exchange(self, x, overlap: bool = False):
        """Halo entries of ``x`` from the left neighbour, mine to the right.  ``overlap`` (own kernels only): the
        asynchronous form — evaluation calls that do not read a halo entry overlap it (``iem_halo_exchange_async``)."""
        if self.kind == "own":
            return self.gm.halo_exchange_async(x) if overlap else self.gm.halo_exchange(x)
        b, dist, ops = self._dev(x), self.dist, []
        # gloo moves host memory: device tensors are staged through the host (the one-GPU rehearsal of the fallback)
        stage = x.is_cuda and dist.get_backend() == "gloo"
        send = recv = None
        if self._right is not None:
            b["send"].copy_(x[b["src"]])
            send = b["send"].cpu() if stage else b["send"]
            ops.append(dist.P2POp(dist.isend, send, self._right))
        if self._left is not None:
            recv = self._torch.empty(b["recv"].shape, dtype=b["recv"].dtype) if stage else b["recv"]
            ops.append(dist.P2POp(dist.irecv, recv, self._left))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        if self._left is not None:
            x[b["dst"]] = recv.to(x.device) if stage else recv
        return x

    def allreduce_obj_grad(self, obj_dev, g):
        """Sum of the scalar objective (1-element tensor, may be ``None``) and of the replicated entries of ``g`` over
        the ranks, in place.  Own kernels: rank-order sum, identical bits on every rank; fallback: the collective's."""
        if self.kind == "own":
            return self.gm.allreduce_obj_grad(obj_dev, g)
        ref = g if g is not None else obj_dev
        b = self._dev(ref)
        if g is None and self._n_shared:
            raise ValueError("null gradient but the model has replicated variables")
        zero = self._torch.zeros(1, dtype=self._torch.float64, device=ref.device)
        out = allreduce_obj_grad_device(obj_dev if obj_dev is not None else zero, g, b["shared"], b["red"], self.dist)
        if obj_dev is not None:
            obj_dev.copy_(out)
        return obj_dev, g


def allreduce_obj_grad_device(obj_dev, grad_dev, shared_idx_dev, buf, dist=None):
    """Device-resident form of :func:`allreduce_obj_grad` (no host round trip, stream-ordered):
    ``obj_dev`` is a 1-element tensor (``ExaModel.obj_device``), ``buf`` a preallocated
    ``1 + len(shared)`` float64 tensor.  Returns ``buf[0:1]`` (the global objective, on device)."""
    if dist is None:
        import torch.distributed as dist
    buf[0:1] = obj_dev
    if shared_idx_dev.numel():
        buf[1:] = grad_dev[shared_idx_dev]
    dist.all_reduce(buf)
    if shared_idx_dev.numel():
        grad_dev[shared_idx_dev] = buf[1:]
    return buf[0:1]


def allreduce_obj_grad(obj_local: float, grad_local, shared_idx, dist=None):
    """The only data-path collective: sum the scalar objective and the gradient entries of
    replicated variables in ONE small buffer (8·(1+len(shared_idx)) bytes)."""
    import torch
    if dist is None:
        import torch.distributed as dist
    buf = torch.empty(1 + len(shared_idx), dtype=torch.float64, device=grad_local.device)
    buf[0] = obj_local
    if len(shared_idx):
        buf[1:] = grad_local[shared_idx]
    dist.all_reduce(buf)
    if len(shared_idx):
        grad_local[shared_idx] = buf[1:]
    return float(buf[0].item()), grad_local
