"""``ExaModel``: the NLPModels evaluation surface over ``libiem_hip.so``.

Mirror of what the reference obtains from ``ExaModels.ExaModel(core)``
(``/root/reference/src/infiniteopt_backend.jl:156``) and what the solvers call on it
every iteration (``ext/InfiniteExaModelsIpopt.jl:49``, ``ext/InfiniteExaModelsMadNLP.jl:50``):
``obj, grad!, cons!, jac_structure!, jac_coord!, hess_structure!, hess_coord!``, the
``meta`` block (``nvar, ncon, nnzj, nnzh, x0, y0, lvar, uvar, lcon, ucon, minimize``) and
``θ``.  Vectors are torch float64 tensors resident in HBM (the ROCArray analogue);
only raw device pointers cross the C-ABI.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import Optional

import numpy as np

from . import lib as _lib
from .core import ExaCore


class MI355XBackend:
    """Value for ``backend =`` that selects the HIP evaluator (the reference's
    ``CUDABackend()`` slot, ``README.md:41``)."""

    def __init__(self, device: int = 0, shard=None):
        """``shard = (group, rank, world)``: this process is rank ``rank`` of a ``world``-GPU run sharded over
        the supports of infinite-parameter group ``group`` (1 = the first parameter): the backend still
        transcribes the GLOBAL model, ``iem_create_sharded`` cuts the rank's window."""
        self.device = int(device)
        self.shard = tuple(int(v) for v in shard) if shard is not None else None

    def __repr__(self):
        return f"MI355XBackend(device={self.device}" + (f", shard={self.shard})" if self.shard else ")")


KERNEL_KINDS = ("cons", "jac", "hess", "obj", "grad", "jprod", "jtprod", "hprod", "pair", "trial", "accepted", "point")   # iem_kernel_info_t.kind


def _ptr(t) -> int:
    if t is None:
        return 0
    if hasattr(t, "data_ptr"):
        return int(t.data_ptr())
    return int(t)


class ExaModel:
    """Device-resident NLP model built from an :class:`ExaCore`."""

    @classmethod
    def from_blob(cls, blob: bytes, device: int = 0, hess_layout: str = "exa", options: Optional[dict] = None) -> "ExaModel":
        """Model from a blob written by any producer of include/iem_blob.h (e.g. the Julia
        writer in julia/MI355XBackend.jl) — no host-side core."""
        return cls(None, device=device, blob=blob, hess_layout=hess_layout, options=options)

    @classmethod
    def sharded(cls, blob: bytes, group: int, rank: int, world: int, device: int = 0, hess_layout: str = "exa",
                options: Optional[dict] = None) -> "ExaModel":
        """Rank ``rank``'s shard of the GLOBAL ``blob`` (``iem_create_sharded``): the window over the
        supports of infinite-parameter group ``group`` is cut inside the library.  ``shard_info()``,
        ``shard_var_map()`` and ``shard_templates()`` say what this rank holds; ``comm_export`` /
        ``comm_connect`` wire the ranks' mailboxes for ``halo_exchange`` and ``allreduce_obj_grad``."""
        return cls(None, device=device, blob=blob, hess_layout=hess_layout, options=options, _shard=(int(group), int(rank), int(world)))

    def __init__(self, core: Optional[ExaCore], device: Optional[int] = None, blob: Optional[bytes] = None,
                 hess_layout: str = "exa", options: Optional[dict] = None, _shard=None):
        """``hess_layout``: ``"exa"`` = ExaModels' COO layout (default, what parity is stated on);
        ``"merged"`` = opt-in layout in which duplicate ``(row, col)`` slots of one support are
        summed in registers (smaller ``nnzh``; ``hess_structure``/``hess_coord`` stay consistent).
        ``options``: generator knobs for THIS handle only (``iem_create_opts``); the process
        defaults (``lib.set_option`` / ``lib.options``) apply to whatever is not named."""
        import torch

        if device is None:
            device = core.backend.device if core is not None and isinstance(core.backend, MI355XBackend) else 0
        if not torch.cuda.is_available():
            raise _lib.IemError("no GPU visible: the evaluation path has no CPU fallback")
        self._torch = torch
        self.core = core
        self.device = torch.device("cuda", device)
        self._L = _lib.lib()
        blob = blob if blob is not None else core.to_blob()
        if hess_layout not in ("exa", "merged"):
            raise ValueError(hess_layout)
        self.hess_layout = hess_layout
        h = C.c_void_p()
        hopts = dict(options or {})
        hopts["hess_merge"] = 1 if hess_layout == "merged" else 0
        arr, n = _lib.option_array(hopts)
        if _shard is None:
            _lib.check(self._L.iem_create_opts(blob, len(blob), device, arr, n, C.byref(h)))
        else:
            _lib.check(self._L.iem_create_sharded(blob, len(blob), device, _shard[0], _shard[1], _shard[2], arr, n, C.byref(h)))
        self._h = h
        if core is not None:
            core._model = self
        m = _lib.Meta()
        _lib.check(self._L.iem_meta(self._h, C.byref(m)))
        host = lambda which, n: self._host(which, n)
        self.meta = SimpleNamespace(
            nvar=int(m.nvar), ncon=int(m.ncon), npar=int(m.npar), nnzj=int(m.nnzj), nnzh=int(m.nnzh),
            minimize=bool(m.minimize), n_templates=int(m.n_templates), n_kernels=int(m.n_kernels),
            x0=host(0, m.nvar), lvar=host(1, m.nvar), uvar=host(2, m.nvar),
            lcon=host(3, m.ncon), ucon=host(4, m.ncon), y0=host(5, m.ncon))
        self.counters = SimpleNamespace(neval_obj=0, neval_grad=0, neval_cons=0, neval_jac=0, neval_hess=0)

    # ---- lifecycle -----------------------------------------------------------
    def close(self):
        cached = self.__dict__.pop("_newton_linear", None)     # what a solver keeps on the model between solves (contrib/newton.py)
        if cached is not None:
            cached[0].close()
        if getattr(self, "_h", None):
            self._L.iem_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _host(self, which: int, n: int) -> np.ndarray:
        out = np.zeros(max(int(n), 1))
        _lib.check(self._L.iem_get_host(self._h, which, out.ctypes.data))
        return out[:int(n)]

    @property
    def theta(self) -> np.ndarray:
        """``model.θ`` (``infiniteopt_backend.jl:479``) — host mirror."""
        return self._host(6, self.meta.npar)

    def template_info(self, i: int) -> dict:
        t = _lib.TemplateInfo()
        _lib.check(self._L.iem_template_info(self._h, i, C.byref(t)))
        return {k: int(getattr(t, k)) for k, _ in t._fields_}

    def kernels(self):
        """Launch shape and algorithmic traffic of every fused kernel of this model."""
        out = []
        for k in range(self.meta.n_kernels):
            ki = _lib.KernelInfo()
            _lib.check(self._L.iem_kernel_info(self._h, k, C.byref(ki)))
            out.append(dict(name=ki.name.decode(), kind=KERNEL_KINDS[ki.kind],
                            grid=tuple(ki.grid), lds_bytes=int(ki.lds_bytes), jit=bool(ki.jit),
                            alg_bytes_read=int(ki.alg_bytes_read), alg_bytes_written=int(ki.alg_bytes_written)))
        return out

    def set_parameter(self, offset: int, vals) -> None:
        v = np.ascontiguousarray(vals, dtype=np.float64)
        _lib.check(self._L.iem_set_parameter(self._h, int(offset), v.shape[0], v.ctypes.data))

    def _sync_stream(self):
        s = self._torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._L.iem_set_stream(self._h, C.c_void_p(s)))

    def _new(self, n: int):
        return self._torch.empty(int(n), dtype=self._torch.float64, device=self.device)

    def _chk(self, t, n: int, name: str):
        if t.dtype != self._torch.float64 or not t.is_cuda or not t.is_contiguous() or t.numel() != n:
            raise ValueError(f"{name} must be a contiguous float64 CUDA tensor of length {n}")

    # ---- NLPModels API ---------------------------------------------------------
    def obj(self, x) -> float:
        self._chk(x, self.meta.nvar, "x")
        self._sync_stream()
        out = C.c_double()
        _lib.check(self._L.iem_obj(self._h, _ptr(x), C.byref(out)))
        self.counters.neval_obj += 1
        return float(out.value)

    def obj_device(self, x, out=None):
        """Asynchronous ``obj``: the scalar stays on the device (graph-capturable)."""
        self._chk(x, self.meta.nvar, "x")
        out = out if out is not None else self._new(1)
        self._sync_stream()
        _lib.check(self._L.iem_obj_device(self._h, _ptr(x), _ptr(out)))
        self.counters.neval_obj += 1
        return out

    def obj_begin(self, x) -> None:
        """First half of ``obj``: enqueue the objective kernel and return (``iem_obj_begin``).  A caller that evaluates
        obj, grad!, cons!, jac_coord!, hess_coord! at one point calls this first and :meth:`obj_end` after its last
        launch — the scalar's host round trip overlaps the other four calls."""
        self._chk(x, self.meta.nvar, "x")
        self._sync_stream()
        _lib.check(self._L.iem_obj_begin(self._h, _ptr(x)))
        self.counters.neval_obj += 1

    def obj_end(self) -> float:
        out = C.c_double()
        _lib.check(self._L.iem_obj_end(self._h, C.byref(out)))
        return float(out.value)

    def grad(self, x, g=None):
        """``grad!(m, x, g)``."""
        self._chk(x, self.meta.nvar, "x")
        g = g if g is not None else self._new(self.meta.nvar)
        self._chk(g, self.meta.nvar, "g")
        self._sync_stream()
        _lib.check(self._L.iem_grad(self._h, _ptr(x), _ptr(g)))
        self.counters.neval_grad += 1
        return g

    def cons(self, x, c=None):
        """``cons!(m, x, c)``."""
        self._chk(x, self.meta.nvar, "x")
        c = c if c is not None else self._new(self.meta.ncon)
        self._chk(c, self.meta.ncon, "c")
        self._sync_stream()
        _lib.check(self._L.iem_cons(self._h, _ptr(x), _ptr(c)))
        self.counters.neval_cons += 1
        return c

    def jac_coord(self, x, vals=None):
        """``jac_coord!(m, x, vals)``."""
        self._chk(x, self.meta.nvar, "x")
        vals = vals if vals is not None else self._new(self.meta.nnzj)
        self._chk(vals, self.meta.nnzj, "vals")
        self._sync_stream()
        _lib.check(self._L.iem_jac_coord(self._h, _ptr(x), _ptr(vals)))
        self.counters.neval_jac += 1
        return vals

    def hess_coord(self, x, y, vals=None, obj_weight: float = 1.0):
        """``hess_coord!(m, x, y, vals; obj_weight)``."""
        self._chk(x, self.meta.nvar, "x")
        self._chk(y, self.meta.ncon, "y")
        vals = vals if vals is not None else self._new(self.meta.nnzh)
        self._chk(vals, self.meta.nnzh, "vals")
        self._sync_stream()
        _lib.check(self._L.iem_hess_coord(self._h, _ptr(x), _ptr(y), float(obj_weight), _ptr(vals)))
        self.counters.neval_hess += 1
        return vals

    def jac_hess_coord(self, x, y, jac=None, hess=None, obj_weight: float = 1.0):
        """``jac_coord!(m, x, jac)`` and ``hess_coord!(m, x, y, hess; obj_weight)`` in ONE launch
        (``iem_jac_hess_coord``): identical bytes to the two calls."""
        self._chk(x, self.meta.nvar, "x"); self._chk(y, self.meta.ncon, "y")
        jac = jac if jac is not None else self._new(self.meta.nnzj)
        hess = hess if hess is not None else self._new(self.meta.nnzh)
        self._chk(jac, self.meta.nnzj, "jac"); self._chk(hess, self.meta.nnzh, "hess")
        self._sync_stream()
        _lib.check(self._L.iem_jac_hess_coord(self._h, _ptr(x), _ptr(y), float(obj_weight), _ptr(jac), _ptr(hess)))
        self.counters.neval_jac += 1
        self.counters.neval_hess += 1
        return jac, hess

    def eval_trial(self, x, c=None, defer_obj: bool = False):
        """``obj(m, x)`` and ``cons!(m, x, c)`` in ONE launch (``iem_eval_trial``: what a line search evaluates at a trial
        point, ``ext/InfiniteExaModelsMadNLP.jl:49-50`` of the reference).  Returns ``(f, c)``; ``defer_obj``: returns
        ``(None, c)`` at once and :meth:`obj_end` collects the value later."""
        self._chk(x, self.meta.nvar, "x")
        c = c if c is not None else self._new(self.meta.ncon)
        self._chk(c, self.meta.ncon, "c")
        self._sync_stream()
        out = C.c_double()
        _lib.check(self._L.iem_eval_trial(self._h, _ptr(x), _ptr(c), None if defer_obj else C.byref(out)))
        self.counters.neval_obj += 1
        self.counters.neval_cons += 1
        return (None if defer_obj else float(out.value)), c

    def eval_accepted(self, x, y, g=None, jac=None, hess=None, obj_weight: float = 1.0):
        """``grad!``, ``jac_coord!`` and ``hess_coord!`` in ONE launch (``iem_eval_accepted``: what a solver evaluates once
        per accepted point, ``ext/InfiniteExaModelsMadNLP.jl:64``).  Identical bytes to the three calls."""
        self._chk(x, self.meta.nvar, "x"); self._chk(y, self.meta.ncon, "y")
        g = g if g is not None else self._new(self.meta.nvar)
        jac = jac if jac is not None else self._new(self.meta.nnzj)
        hess = hess if hess is not None else self._new(self.meta.nnzh)
        self._chk(g, self.meta.nvar, "g"); self._chk(jac, self.meta.nnzj, "jac"); self._chk(hess, self.meta.nnzh, "hess")
        self._sync_stream()
        _lib.check(self._L.iem_eval_accepted(self._h, _ptr(x), _ptr(y), float(obj_weight), _ptr(g), _ptr(jac), _ptr(hess)))
        self.counters.neval_grad += 1
        self.counters.neval_jac += 1
        self.counters.neval_hess += 1
        return g, jac, hess

    def eval_all(self, x, y, c=None, g=None, jac=None, hess=None, obj_weight: float = 1.0, defer_obj: bool = False):
        """All five evaluations of one point in ONE launch (``iem_eval_all``).  Returns ``(f, c, g, jac, hess)``."""
        self._chk(x, self.meta.nvar, "x"); self._chk(y, self.meta.ncon, "y")
        c = c if c is not None else self._new(self.meta.ncon)
        g = g if g is not None else self._new(self.meta.nvar)
        jac = jac if jac is not None else self._new(self.meta.nnzj)
        hess = hess if hess is not None else self._new(self.meta.nnzh)
        self._chk(c, self.meta.ncon, "c"); self._chk(g, self.meta.nvar, "g"); self._chk(jac, self.meta.nnzj, "jac"); self._chk(hess, self.meta.nnzh, "hess")
        self._sync_stream()
        out = C.c_double()
        _lib.check(self._L.iem_eval_all(self._h, _ptr(x), _ptr(y), float(obj_weight), _ptr(c), _ptr(g), _ptr(jac), _ptr(hess),
                                        None if defer_obj else C.byref(out)))
        for k in ("obj", "cons", "grad", "jac", "hess"):
            setattr(self.counters, "neval_" + k, getattr(self.counters, "neval_" + k) + 1)
        return (None if defer_obj else float(out.value)), c, g, jac, hess

    def jprod(self, x, v, Jv=None):
        """``jprod!(m, x, v, Jv)``: Jacobian–vector product (ncon)."""
        self._chk(x, self.meta.nvar, "x"); self._chk(v, self.meta.nvar, "v")
        Jv = Jv if Jv is not None else self._new(self.meta.ncon)
        self._chk(Jv, self.meta.ncon, "Jv")
        self._sync_stream()
        _lib.check(self._L.iem_jprod(self._h, _ptr(x), _ptr(v), _ptr(Jv)))
        return Jv

    def jtprod(self, x, v, Jtv=None):
        """``jtprod!(m, x, v, Jtv)``: transposed Jacobian–vector product (nvar)."""
        self._chk(x, self.meta.nvar, "x"); self._chk(v, self.meta.ncon, "v")
        Jtv = Jtv if Jtv is not None else self._new(self.meta.nvar)
        self._chk(Jtv, self.meta.nvar, "Jtv")
        self._sync_stream()
        _lib.check(self._L.iem_jtprod(self._h, _ptr(x), _ptr(v), _ptr(Jtv)))
        return Jtv

    def hprod(self, x, y, v, Hv=None, obj_weight: float = 1.0):
        """``hprod!(m, x, y, v, Hv; obj_weight)``: Lagrangian-Hessian–vector product (nvar)."""
        self._chk(x, self.meta.nvar, "x"); self._chk(y, self.meta.ncon, "y"); self._chk(v, self.meta.nvar, "v")
        Hv = Hv if Hv is not None else self._new(self.meta.nvar)
        self._chk(Hv, self.meta.nvar, "Hv")
        self._sync_stream()
        _lib.check(self._L.iem_hprod(self._h, _ptr(x), _ptr(y), _ptr(v), float(obj_weight), _ptr(Hv)))
        return Hv

    def jac_structure(self, base: int = 0):
        """``jac_structure!(m, rows, cols)`` → host int64 arrays (``base`` 1 = Julia)."""
        r = np.zeros(max(self.meta.nnzj, 1), dtype=np.int64)
        c = np.zeros(max(self.meta.nnzj, 1), dtype=np.int64)
        _lib.check(self._L.iem_jac_structure(self._h, r.ctypes.data, c.ctypes.data, base))
        return r[:self.meta.nnzj], c[:self.meta.nnzj]

    def hess_structure(self, base: int = 0):
        r = np.zeros(max(self.meta.nnzh, 1), dtype=np.int64)
        c = np.zeros(max(self.meta.nnzh, 1), dtype=np.int64)
        _lib.check(self._L.iem_hess_structure(self._h, r.ctypes.data, c.ctypes.data, base))
        return r[:self.meta.nnzh], c[:self.meta.nnzh]

    def jac_structure_device(self, base: int = 0):
        """Structure generated on the device (int64 tensors), for GPU-resident solvers."""
        t = self._torch
        r = t.empty(self.meta.nnzj, dtype=t.int64, device=self.device)
        c = t.empty(self.meta.nnzj, dtype=t.int64, device=self.device)
        self._sync_stream()
        _lib.check(self._L.iem_jac_structure_device(self._h, _ptr(r), _ptr(c), base))
        return r, c

    def hess_structure_device(self, base: int = 0):
        t = self._torch
        r = t.empty(self.meta.nnzh, dtype=t.int64, device=self.device)
        c = t.empty(self.meta.nnzh, dtype=t.int64, device=self.device)
        self._sync_stream()
        _lib.check(self._L.iem_hess_structure_device(self._h, _ptr(r), _ptr(c), base))
        return r, c

    def synchronize(self):
        _lib.check(self._L.iem_synchronize(self._h))

    def raw_pair(self, x, y, jac, hess, obj_weight: float = 1.0, fused: bool = False, halo: bool = False):
        """``step()`` closure for hot loops: one ``iem_jac_coord`` + one ``iem_hess_coord`` on fixed buffers with the
        Python-side argument checks and the stream lookup done ONCE here (what a compiled host — the Julia
        ``ccall`` shim — pays per call is the two C calls, not ~10 µs of interpreter work each)."""
        self._chk(x, self.meta.nvar, "x"); self._chk(y, self.meta.ncon, "y")
        self._chk(jac, self.meta.nnzj, "vals"); self._chk(hess, self.meta.nnzh, "vals")
        self._sync_stream()
        L, h = self._L, self._h
        px, py, pj, ph, w = _ptr(x), _ptr(y), _ptr(jac), _ptr(hess), float(obj_weight)
        jac_coord, hess_coord, check = L.iem_jac_coord, L.iem_hess_coord, _lib.check
        pair, halo_async = L.iem_jac_hess_coord, L.iem_halo_exchange_async

        # `fused`: the one-launch form (iem_jac_hess_coord); `halo`: every step first starts the asynchronous halo
        # exchange of x (what a solver iteration, which moves x, adds around the pair on a sharded handle)
        if fused and halo:
            def step():
                rc = halo_async(h, px) or pair(h, px, py, w, pj, ph)
                if rc:
                    check(rc)
        elif fused:
            def step():
                rc = pair(h, px, py, w, pj, ph)
                if rc:
                    check(rc)
        elif halo:
            def step():
                rc = halo_async(h, px) or jac_coord(h, px, pj) or hess_coord(h, px, py, w, ph)
                if rc:
                    check(rc)
        else:
            def step():
                rc = jac_coord(h, px, pj) or hess_coord(h, px, py, w, ph)
                if rc:
                    check(rc)
        return step

    def raw_loop(self, x, y, g, c, jac, hess, obj_weight: float = 1.0, fused: bool = True, defer_obj: bool = True, phases: bool = False, one_launch: bool = False):
        """``step() -> f`` closure: the five evaluations a solver makes at one point — obj, grad!, cons!, jac_coord!,
        hess_coord! (``ext/InfiniteExaModelsIpopt.jl:48-49``) — on fixed buffers, argument checks and stream lookup done
        once.  ``defer_obj``: the objective is launched FIRST (``iem_obj_begin``) and its value collected LAST
        (``iem_obj_end``), so the scalar's host round trip overlaps the other four calls instead of stalling the host in
        front of them; ``fused``: jac_coord! + hess_coord! through the one-launch form."""
        self._chk(x, self.meta.nvar, "x"); self._chk(y, self.meta.ncon, "y"); self._chk(g, self.meta.nvar, "g")
        self._chk(c, self.meta.ncon, "c"); self._chk(jac, self.meta.nnzj, "vals"); self._chk(hess, self.meta.nnzh, "vals")
        self._sync_stream()
        L, h, check = self._L, self._h, _lib.check
        px, py, pg, pc, pj, ph, w = _ptr(x), _ptr(y), _ptr(g), _ptr(c), _ptr(jac), _ptr(hess), float(obj_weight)
        out = C.c_double()
        ref = C.byref(out)
        obj, begin, end, grad, cons = L.iem_obj, L.iem_obj_begin, L.iem_obj_end, L.iem_grad, L.iem_cons
        jacf, hessf, pair = L.iem_jac_coord, L.iem_hess_coord, L.iem_jac_hess_coord

        trial, accepted, allf = L.iem_eval_trial, L.iem_eval_accepted, L.iem_eval_all
        if one_launch:  # all five evaluations of the point in ONE launch
            def step():
                rc = allf(h, px, py, w, pc, pg, pj, ph, ref)
                if rc:
                    check(rc)
                return out.value
            return step
        if phases:      # one launch per solver phase: obj + cons! (value collected last), then grad! + jac_coord! + hess_coord!
            def step():
                rc = trial(h, px, pc, None) or accepted(h, px, py, w, pg, pj, ph) or end(h, ref)
                if rc:
                    check(rc)
                return out.value
            return step

        def step():
            rc = (begin(h, px) if defer_obj else obj(h, px, ref)) or grad(h, px, pg) or cons(h, px, pc) or \
                (pair(h, px, py, w, pj, ph) if fused else (jacf(h, px, pj) or hessf(h, px, py, w, ph))) or \
                (end(h, ref) if defer_obj else 0)
            if rc:
                check(rc)
            return out.value
        return step

    # ---- sharding / multi-GPU ------------------------------------------------------
    def shard_info(self) -> dict:
        t = _lib.ShardT()
        _lib.check(self._L.iem_shard_info(self._h, C.byref(t)))
        return t.asdict()

    def shard_var_map(self):
        """``(local -> global variable, flags)``; flag bit 0 owned here, bit 1 replicated, bit 2 halo copy."""
        vm = np.zeros(max(self.meta.nvar, 1), dtype=np.int64)
        vf = np.zeros(max(self.meta.nvar, 1), dtype=np.uint8)
        _lib.check(self._L.iem_shard_var_map(self._h, vm.ctypes.data, vf.ctypes.data))
        return vm[:self.meta.nvar], vf[:self.meta.nvar]

    def shard_templates(self):
        """Per local template: where it sits in the global model; ``["ordinals"]`` = global item ordinal of every
        local item (global row = ``global_o0 + ordinal``, COO position = ``global_o1 + o1step·ordinal + slot``)."""
        n = C.c_int64()
        _lib.check(self._L.iem_shard_template_items(self._h, None, C.byref(n)))
        items = np.zeros(max(int(n.value), 1), dtype=np.int64)
        _lib.check(self._L.iem_shard_template_items(self._h, items.ctypes.data, C.byref(n)))
        out = []
        for i in range(self.meta.n_templates):
            t = _lib.ShardTemplate()
            _lib.check(self._L.iem_shard_template_info(self._h, i, C.byref(t)))
            out.append(t.asdict(items))
        return out

    def comm_export(self) -> bytes:
        buf = C.create_string_buffer(_lib.COMM_HANDLE_BYTES)
        _lib.check(self._L.iem_comm_export(self._h, buf))
        return buf.raw

    def comm_connect(self, handles: bytes) -> None:
        """``handles``: every rank's ``comm_export()`` concatenated in rank order."""
        _lib.check(self._L.iem_comm_connect(self._h, handles))

    def halo_exchange(self, x):
        """Fill the halo entries of the local ``x`` from the left neighbour (and send mine right);
        asynchronous on the current stream."""
        self._chk(x, self.meta.nvar, "x")
        self._sync_stream()
        _lib.check(self._L.iem_halo_exchange(self._h, _ptr(x)))
        return x

    def halo_exchange_async(self, x):
        """The same exchange off the critical path (``iem_halo_exchange_async``): nothing is launched — the exchange rides
        on the first evaluation launch that takes ``x`` and cannot touch a halo entry of it (one extra leading workgroup
        of that kernel); a call that can touch one gets the stand-alone exchange in front of it.  Nothing else may write
        ``x`` or read its halo entries until that call (or :meth:`halo_wait`)."""
        self._chk(x, self.meta.nvar, "x")
        self._sync_stream()
        _lib.check(self._L.iem_halo_exchange_async(self._h, _ptr(x)))
        return x

    def halo_wait(self) -> None:
        self._sync_stream()
        _lib.check(self._L.iem_halo_wait(self._h))

    def halo_reads(self) -> dict:
        """Per kernel kind ``(touches a halo entry through x, through a variable-space v, can carry a deferred exchange)``."""
        out = {}
        for k, name in enumerate(KERNEL_KINDS):
            a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
            _lib.check(self._L.iem_halo_reads(self._h, k, C.byref(a), C.byref(b), C.byref(c)))
            out[name] = (bool(a.value), bool(b.value), bool(c.value))
        return out

    def halo_fold(self, vec):
        """Transposed halo exchange for a variable-space vector (``jtprod`` of this rank's rows): halo-copy
        entries are added to the left neighbour's owned entries and zeroed here; asynchronous."""
        self._chk(vec, self.meta.nvar, "vec")
        self._sync_stream()
        _lib.check(self._L.iem_halo_fold(self._h, _ptr(vec)))
        return vec

    def allreduce_obj_grad(self, obj_dev, g):
        """Sum the device scalar ``obj_dev`` and the replicated entries of ``g`` over the ranks, in place."""
        if g is not None:
            self._chk(g, self.meta.nvar, "g")
        self._sync_stream()
        _lib.check(self._L.iem_allreduce_obj_grad(self._h, _ptr(obj_dev), _ptr(g)))
        return obj_dev, g

    def comm_status(self) -> int:
        st = C.c_int64()
        _lib.check(self._L.iem_comm_status(self._h, C.byref(st)))
        return int(st.value)

    def tune(self, x, y, jac=None, hess=None, obj_weight: float = 1.0) -> dict:
        """Decide the store-batch variant for these output buffers now (``iem_tune``: twenty-one complete
        evaluations into each, then a stream synchronise) instead of during the first calls of a solve."""
        self._chk(x, self.meta.nvar, "x")
        if hess is not None:
            self._chk(y, self.meta.ncon, "y"); self._chk(hess, self.meta.nnzh, "hess")
        if jac is not None:
            self._chk(jac, self.meta.nnzj, "jac")
        self._sync_stream()
        _lib.check(self._L.iem_tune(self._h, _ptr(x), _ptr(y) if hess is not None else None, float(obj_weight),
                                    _ptr(jac) if jac is not None else None, _ptr(hess) if hess is not None else None))
        return {"jac": self.tuner_choice("jac", jac) if jac is not None else -1,
                "hess": self.tuner_choice("hess", hess) if hess is not None else -1}

    def tuner_choice(self, kind: str, out) -> int:
        """Which code object ``jac_coord!`` / ``hess_coord!`` (kind "jac" / "hess") uses for output
        buffer ``out``: -1 still measuring or tuner off, 0 default, 1 large store batch."""
        ch = C.c_int32(-1)
        _lib.check(self._L.iem_tuner_choice(self._h, {"jac": 0, "hess": 1}[kind], _ptr(out), C.byref(ch)))
        return int(ch.value)

    def time_kernels(self, x, y, jac, hess, iters: int = 20):
        """Average device time (ms) of one jac_coord! and one hess_coord! call, HIP events
        on the launch stream."""
        self._sync_stream()
        a, b = C.c_double(), C.c_double()
        _lib.check(self._L.iem_time_kernels(self._h, _ptr(x), _ptr(y), _ptr(jac), _ptr(hess), iters,
                                            C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)
