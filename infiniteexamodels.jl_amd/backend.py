"""``ExaTranscriptionBackend``: the plug point of the reference
(``/root/reference/src/infiniteopt_backend.jl:85-157``).

Mirrored: the ``backend`` slot that selects the device evaluator (``:100``, forwarded at ``:155``),
``build_transformation_backend!`` (``:150-157``), ``empty!`` (``:134-143``), solver attributes (``:160-252``), ``optimize!``
(``:259-271``), the object / support mappings (``:274-350``), result queries (``:352-508``: statuses, objective,
``map_value``, ``map_dual``), the parameter / start-value update hooks (``:511-592``) and warm starts (``:595-615``).
A solver is any callable ``solver(model, x0, y0, **options) -> result`` with ``result.solution`` / ``result.multipliers``
(torch or numpy vectors; ``contrib/results.py`` lists the optional fields); one that keeps state between solves may offer
``resolve(model, x0, y0, **changed_options)`` and is then handed only the options that changed, as the reference's
extensions do (``ext/InfiniteExaModelsIpopt.jl:10-60``).
"""
from __future__ import annotations

import time
from typing import Any, Callable, Dict, Optional

import numpy as np

from . import transcribe
from .core import ExaCore
from .infinite import FiniteParameterRef, InfiniteModel, ParameterFunctionRef
from .model import ExaModel, MI355XBackend


class _LazyResults:
    """``contrib.results`` (status tables, result objects: outside SURVEY §8's scope, see ``contrib/__init__.py``) is only
    loaded when a result query or an ``optimize()`` needs it; the evaluation path never imports it."""

    def __getattr__(self, name):
        from .contrib import results
        return getattr(results, name)


_results = _LazyResults()


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
        self._solved_once = False
        self._y0 = None          # a rebuild discards the warm start with the model (src/infiniteopt_backend.jl:595-615 keeps it inside the model)
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

    # solver settings (:160-252) ------------------------------------------------------------
    def get_attribute(self, attr: str):
        """``JuMP.get_attribute(backend, attr)``: a raw solver option, or ``"silent"`` / ``"time_limit_sec"`` /
        ``"solver_name"`` (the MOI attributes of ``:187-231``)."""
        if attr == "silent":
            return self.silent
        if attr == "time_limit_sec":
            return None if np.isnan(self.time_limit) else self.time_limit
        if attr == "solver_name":
            s = self.solver
            return "No solver attached" if s is None else getattr(s, "__name__", type(s).__name__)
        if attr not in self.options:
            raise KeyError(f"Attribute `{attr}` not found.")
        return self.options[attr]

    def set_attribute(self, attr: str, value) -> None:
        if attr == "silent":
            self.silent = bool(value)
        elif attr == "time_limit_sec":
            self.time_limit = float("nan") if value is None else float(value)
        else:
            self.solve_time = float("nan")
            self.options[attr] = value

    def set_silent(self, value: bool = True) -> None:
        self.set_attribute("silent", value)

    def set_time_limit_sec(self, value) -> None:
        self.set_attribute("time_limit_sec", value)

    def set_optimizer(self, solver, **params) -> None:
        """``JuMP.set_optimizer(backend, solver_type)`` (``:233-252``): previous solver-specific settings are dropped."""
        self.options.clear()
        self.prev_options = {}
        self.solver = solver
        self._solved_once = False
        for k, v in params.items():
            self.set_attribute(k, v)

    # JuMP.optimize!(backend) (:259-271)
    def optimize(self):
        self._refuse_sharded("optimize")
        if self.solver is None:
            raise RuntimeError("No solver attached to the backend")      # JuMP.NoOptimizer()
        if self.core is None:
            self.build_transformation_backend()
        t0 = time.perf_counter()
        y0 = getattr(self, "_y0", None)
        if y0 is None:      # (backend = None builds the core only: a host-side solver evaluates it its own way)
            y0 = self.model.meta.y0 if self.model is not None else np.zeros(self.core.ncon)
        conv = getattr(self.solver, "option_convention", None)
        if isinstance(conv, str):
            conv = _results.OPTION_CONVENTIONS[conv]
        if conv is not None:      # silent / time limit become the solver's own options; only changes are new
            changed = _results.process_options(self.options, self, conv)
        else:
            changed = {k: v for k, v in self.options.items() if k not in self.prev_options or self.prev_options[k] != v}
            self.prev_options.update(changed)
        if getattr(self, "_solved_once", False) and hasattr(self.solver, "resolve"):
            self.results = self.solver.resolve(self.model, self.core.x0.copy(), y0, **changed)
        else:                     # a callable keeps no state: it sees every option in force
            self.results = self.solver(self.model, self.core.x0.copy(), y0, **self.prev_options)
        self._solved_once = True
        self.solve_time = time.perf_counter() - t0
        return self.results

    # object / support mappings (:274-350) -----------------------------------------------------
    def transformation_constraint(self, cref):
        return self.data.constraint_mappings[cref]

    def _supports(self, group_idxs, label: str):
        """Array of support tuples over the groups' grid: shape ``dims + (number of parameters,)``, filtered by label
        (``"public"``: user-visible supports — internal collocation nodes dropped; ``"all"``; ``"internal"``)."""
        groups = [self._inf_model.groups[g - 1] for g in group_idxs]
        grids = np.meshgrid(*[np.arange(g.num_supports) for g in groups], indexing="ij") if groups else []
        cols = [g.supports[ix] for g, ix in zip(groups, grids)]       # each: dims + (n_prefs of the group,)
        arr = np.concatenate(cols, axis=-1) if cols else np.zeros((0,))
        return self._label_filter(arr, group_idxs, label)

    def _label_filter(self, arr, group_idxs, label: str):
        """``_label_filter`` (``:303-315``): keep the supports carrying ``label`` along every group axis."""
        if label == "all" or not group_idxs:
            return arr
        if label == "public" and not any(self.data.has_internal_supps[g - 1] for g in group_idxs):
            return arr
        arr = np.asarray(arr)
        dims = tuple(self._inf_model.groups[g - 1].num_supports for g in group_idxs)
        if arr.shape[:len(dims)] != dims:      # (a product iterator of the reference is a vector: first group fastest)
            if arr.ndim >= 1 and arr.shape[0] == int(np.prod(dims)):
                arr = arr.reshape(dims + arr.shape[1:], order="F")
            else:
                raise IndexError(f"cannot filter an array of shape {arr.shape} by the supports of groups {list(group_idxs)} (a restricted constraint?)")
        for axis, g in enumerate(group_idxs):
            grp = self._inf_model.groups[g - 1]
            internal = grp.internal if grp.internal is not None else np.zeros(grp.num_supports, dtype=bool)
            keep = internal if label == "internal" else ~internal
            arr = np.compress(keep, arr, axis=axis)
        return arr

    def variable_supports(self, vref, label: str = "public"):
        """``InfiniteOpt.variable_supports`` (``:317-331``)."""
        return self._supports(vref.group_idxs, label)

    def constraint_supports(self, cref, label: str = "public"):
        """``InfiniteOpt.constraint_supports`` (``:333-350``); a domain restriction keeps the supports it admits (a vector)."""
        from .infinite import parameter_group_int_indices
        group_idxs = parameter_group_int_indices(cref.func)
        supps = self._supports(group_idxs, label)
        if cref.restriction is None:
            return supps
        flat = supps.reshape(-1, supps.shape[-1])
        cols = self._restriction_columns(cref.restriction, group_idxs)
        return np.array([s for s in flat if cref.restriction([s[c] for c in cols])])

    def _restriction_columns(self, restriction, group_idxs):
        """columns of a support tuple (over ``group_idxs``) holding the parameters a restriction is written in"""
        order = [p for g in group_idxs for p in self._inf_model.groups[g - 1].prefs]
        return [next(i for i, q in enumerate(order) if q is p) for p in restriction.parameter_refs]

    # result queries (:352-508) -------------------------------------------------------------------
    def _check_results_available(self) -> None:
        if self.results is None:
            raise RuntimeError("No solution available to query.")

    def result_count(self) -> int:
        return 0 if self.results is None else 1

    def raw_status(self) -> str:
        return "optimize not called" if self.results is None else str(self.results.status)

    def termination_status(self) -> str:
        if self.results is None:
            return "OPTIMIZE_NOT_CALLED"
        return _results.translate_termination_status(self.solver, self.results.status)

    def primal_status(self) -> str:
        if self.results is None:
            return "NO_SOLUTION"
        return _results.translate_result_status(self.solver, self.results.status)

    dual_status = primal_status

    def solve_time_sec(self) -> float:
        self._check_results_available()
        return self.solve_time

    def objective_value(self) -> float:
        self._check_results_available()
        return float(self.results.objective)

    def map_value(self, ref, label: str = "public"):
        """``InfiniteOpt.map_value`` (``:448-488``): the solution over a variable's supports (its own shape), a finite
        variable's entry, or — for finite parameters and parameter functions — the slab of θ the model evaluates with."""
        d = self.data
        if ref in d.param_mappings:
            par = d.param_mappings[ref]
            theta = np.asarray(self.core.theta)[par.offset:par.offset + par.length].reshape(par.size, order="F")
            return float(theta.reshape(-1)[0]) if isinstance(ref, FiniteParameterRef) else theta
        self._check_results_available()
        var = self.transformation_variable(ref)
        vals = _results.solution(self.results, var)
        return self._label_filter(vals, getattr(ref, "group_idxs", []) or [], label) if not np.isscalar(vals) else vals

    def map_dual(self, cref, label: str = "public"):
        """``InfiniteOpt.map_dual`` (``:490-508``): JuMP's sign (``-1 x`` the NLPModels multipliers).  Variable-domain
        constraints are the ``(vref, "lower" | "upper" | "fix")`` pairs ``variable(...)`` creates from ``lb / ub / fix``."""
        self._check_results_available()
        if isinstance(cref, tuple):
            vref, kind = cref
            var = self.transformation_variable(vref)
            mL, mU = _results.multipliers_L(self.results, var), _results.multipliers_U(self.results, var)
            d = np.asarray(mL) - np.asarray(mU)
            duals = np.maximum(d, 0.0) if kind == "lower" else np.minimum(d, 0.0) if kind == "upper" else d
            group_idxs = getattr(vref, "group_idxs", []) or []
        else:
            from .infinite import parameter_group_int_indices
            duals = -1.0 * _results.multipliers(self.results, self.transformation_constraint(cref))
            group_idxs = parameter_group_int_indices(cref.func)
            if cref.restriction is not None:
                return duals       # (the admitted supports only: constraint_supports gives them in the same order)
        return self._label_filter(duals, group_idxs, label) if np.ndim(duals) else float(duals)


def _to_numpy(v):
    return v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
