"""``ExaTranscriptionBackend``: the plug point of the reference
(``/root/reference/src/infiniteopt_backend.jl:85-157``).

Only what sits ON the evaluation path is mirrored: the ``backend`` slot that selects the
device evaluator (``:100``, forwarded at ``:155``), ``build_transformation_backend!``
(``:150-157``), ``empty!`` (``:134-143``), the parameter / start-value update hooks
(``:511-592``) and warm starts (``:595-615``).  Solver option diffing, status tables and
result queries are solver glue and are not rebuilt; a solver is
any callable ``solver(model, x0, y0, **options) -> result`` with ``result.solution`` /
``result.multipliers`` (torch or numpy vectors).
"""
from __future__ import annotations

import time
from typing import Any, Callable, Dict, Optional

import numpy as np

from . import transcribe
from .core import ExaCore
from .infinite import FiniteParameterRef, InfiniteModel, ParameterFunctionRef
from .model import ExaModel, MI355XBackend


class ExaTranscriptionBackend:
    """``ExaTranscriptionBackend([solver]; backend = nothing)`` (``:112-131``)."""

    def __init__(self, solver: Optional[Callable] = None, backend: Any = None, **options):
        self.core: Optional[ExaCore] = None
        self.model: Optional[ExaModel] = None
        self.backend = backend
        self.solver = solver
        self.prev_options: Dict[str, Any] = {}
        self.options: Dict[str, Any] = dict(options)
        self.silent = False
        self.time_limit = float("nan")
        self.results = None
        self.solve_time = float("nan")
        self.data = transcribe.ExaMappingData()
        self._inf_model: Optional[InfiniteModel] = None

    def _attach(self, inf_model: InfiniteModel) -> None:
        self._inf_model = inf_model

    # Base.empty!(backend) — options survive (:134-143)
    def empty(self) -> "ExaTranscriptionBackend":
        if self.model is not None:
            self.model.close()
        self.core = self.model = self.results = None
        self.prev_options = {}
        self.solve_time = float("nan")
        self.data = transcribe.ExaMappingData()
        return self

    # InfiniteOpt.build_transformation_backend!(model, backend) (:150-157)
    def build_transformation_backend(self, inf_model: Optional[InfiniteModel] = None) -> None:
        inf_model = inf_model or self._inf_model
        self.empty()
        self._inf_model = inf_model
        self.core = transcribe.exa_core(inf_model, self.data, backend=self.backend)
        if isinstance(self.backend, MI355XBackend) and self.backend.shard is not None:
            # one rank of a multi-GPU run: the same global core on every rank, the window cut in the library;
            # theta stays global there, so update_parameter_value / set_parameter! work unchanged
            self.model = ExaModel.sharded(self.core.to_blob(), *self.backend.shard, device=self.backend.device)
            self.model.core = self.core
            self.core._model = self.model
        elif isinstance(self.backend, MI355XBackend):
            self.model = ExaModel(self.core)            # ExaModels.ExaModel(backend.core)
        elif self.backend is not None:
            raise TypeError(f"unsupported backend {self.backend!r}: use MI355XBackend() (the CPU ExaModel of the "
                            "reference is not rebuilt — there is no CPU evaluation path)")
        inf_model._ready = True

    # transformation_variable / parameter updates -------------------------------------------
    def transformation_variable(self, ref):
        d = self.data
        for table in (d.finvar_mappings, d.infvar_mappings, d.param_mappings):
            if ref in table:
                return table[ref]
        raise KeyError(ref)

    def update_parameter_value(self, pref, value) -> bool:
        """``InfiniteOpt.update_parameter_value`` (:511-550): finite parameters and parameter
        functions update θ in place through ``set_parameter!``; anything else needs a rebuild."""
        if self.core is None or pref not in self.data.param_mappings:
            return False
        par = self.data.param_mappings[pref]
        if isinstance(pref, FiniteParameterRef):
            pref.value = float(value)
            self.core.set_parameter(par, [value])
        elif isinstance(pref, ParameterFunctionRef):
            pref.func = value
            dims = par.size
            vals = transcribe._eval_over_supports(value, self._inf_model, pref.group_idxs, dims)
            self.core.set_parameter(par, vals)
        else:
            return False
        return True

    def update_start_value(self, vref, value) -> bool:
        """``InfiniteOpt.update_variable_info`` start-value branch (:553-592): writes ``core.x0``."""
        if self.core is None:
            return False
        d = self.data
        if vref in d.finvar_mappings:
            self.core.x0[int(d.finvar_mappings[vref].i) - 1] = value
        elif vref in d.infvar_mappings:
            var = d.infvar_mappings[vref]
            if callable(value):
                value = transcribe._eval_over_supports(value, self._inf_model, vref.group_idxs, var.size).reshape(-1, order="F")
            self.core.x0[var.offset:var.offset + var.length] = value
        else:
            return False
        return True

    def warmstart_backend_start_values(self) -> None:
        """``InfiniteOpt.warmstart_backend_start_values`` (:595-615)."""
        self._refuse_sharded("warmstart_backend_start_values")
        if self.results is None:
            import warnings
            warnings.warn("No previous solution values found. Unable to warmstart backend.")
            return
        self.core.x0[:] = np.asarray(_to_numpy(self.results.solution))
        self._y0 = np.asarray(_to_numpy(self.results.multipliers))

    def _refuse_sharded(self, what: str) -> None:
        """A backend built with ``MI355XBackend(shard = (group, rank, world))`` holds ONE rank's shard: its model has
        the rank's local variables (halo copies included), ``core`` the global ones.  It is a build-and-evaluate plug
        point — the evaluation calls, ``halo_exchange`` and ``allreduce_obj_grad`` of ``self.model``, the global θ /
        start-value hooks.  Driving a solver needs one that is itself distributed over the ranks (its KKT solve spans
        the shards); the single-process solver slot of the reference (``:259-271``) cannot do that, so it is refused
        loudly instead of handing a global ``x0`` to a local model."""
        if isinstance(self.backend, MI355XBackend) and self.backend.shard is not None:
            raise NotImplementedError(
                f"{what}: this backend holds rank {self.backend.shard[1]} of {self.backend.shard[2]} of a sharded model "
                "(build-and-evaluate only); solve on an unsharded backend, or drive the shards' evaluation calls "
                "(model.halo_exchange / allreduce_obj_grad) from a distributed solver")

    # JuMP.optimize!(backend) (:259-271)
    def optimize(self):
        self._refuse_sharded("optimize")
        if self.solver is None:
            raise RuntimeError("No solver attached to the backend")
        if self.model is None:
            self.build_transformation_backend()
        t0 = time.perf_counter()
        y0 = getattr(self, "_y0", self.model.meta.y0)
        self.results = self.solver(self.model, self.core.x0.copy(), y0, **self.options)
        self.solve_time = time.perf_counter() - t0
        self.prev_options = dict(self.options)
        return self.results


def _to_numpy(v):
    return v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
