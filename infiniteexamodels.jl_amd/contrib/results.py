"""The result side of the plug point: what the reference reads back from a solver's result object.

``ExaModels.solution / multipliers / multipliers_L / multipliers_U`` as the reference calls them
(``/root/reference/src/infiniteopt_backend.jl:464,500-505``), the JSO-status tables (``:359-392``; MadNLP's own table:
``ext/InfiniteExaModelsMadNLP.jl:67-104``) and the option diffing of a re-solve (``ext/InfiniteExaModelsIpopt.jl:10-39``,
``ext/InfiniteExaModelsMadNLP.jl:11-43``).  Host logic only — nothing here touches the device except to bring a result
vector home once.

A result object is anything with ``solution`` and ``multipliers`` (torch or numpy vectors of length nvar / ncon), and
optionally ``multipliers_L`` / ``multipliers_U`` (nvar), ``objective`` and ``status`` (a JSO symbol as a string:
``"first_order"``, ``"max_iter"`` ...) — ``newton.NewtonResult``, the SciPy wrappers of the tests, or a
``SolverCore.GenericExecutionStats`` seen through a shim.
"""
from __future__ import annotations

from typing import Any, Dict

import numpy as np

from ..core import Constraint, Variable


def _host(v) -> np.ndarray:
    return v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v, dtype=np.float64)


def _slab(vec, var: Variable) -> np.ndarray:
    a = _host(vec)[var.offset:var.offset + var.length]
    return a.reshape(var.size, order="F")          # first index fastest (test/transcription.jl:44-57)


def solution(result, var):
    """``ExaModels.solution(result, var)``: the slab of an infinite variable in its own shape, or the entry of a finite
    one (``var.i`` is 1-based, as in ``src/transform.jl:218``)."""
    if isinstance(var, Variable):
        return _slab(result.solution, var)
    return float(_host(result.solution)[int(var.i) - 1])


def multipliers(result, con: Constraint) -> np.ndarray:
    """``ExaModels.multipliers(result, con)``: the rows of one constraint template, in item order (NLPModels sign)."""
    return _host(result.multipliers)[con.offset:con.offset + con.length].copy()


def _bound_multipliers(result, name: str, var):
    vec = getattr(result, name, None)
    if vec is None:       # a solver that takes no bounds reports none
        return np.zeros(var.size, dtype=np.float64) if isinstance(var, Variable) else 0.0
    if isinstance(var, Variable):
        return _slab(vec, var)
    return float(_host(vec)[int(var.i) - 1])


def multipliers_L(result, var):
    return _bound_multipliers(result, "multipliers_L", var)


def multipliers_U(result, var):
    return _bound_multipliers(result, "multipliers_U", var)


# Standard JSO statuses -> MOI.TerminationStatusCode / MOI.ResultStatusCode (src/infiniteopt_backend.jl:359-382); the codes
# are kept as their MOI names
TERMINATION: Dict[str, str] = {
    "first_order": "LOCALLY_SOLVED", "acceptable": "ALMOST_LOCALLY_SOLVED", "small_step": "SLOW_PROGRESS",
    "infeasible": "INFEASIBLE_OR_UNBOUNDED", "unbounded": "INFEASIBLE_OR_UNBOUNDED", "max_iter": "ITERATION_LIMIT",
    "max_time": "TIME_LIMIT", "user": "INTERRUPTED", "exception": "OTHER_ERROR", "stalled": "OTHER_ERROR",
    "max_eval": "OTHER_LIMIT", "neg_pred": "OTHER_ERROR", "not_desc": "OTHER_ERROR",
}
RESULT: Dict[str, str] = {"first_order": "FEASIBLE_POINT", "acceptable": "NEARLY_FEASIBLE_POINT", "infeasible": "INFEASIBLE_POINT"}
# MadNLP.Status -> MOI (ext/InfiniteExaModelsMadNLP.jl:67-96)
MADNLP_TERMINATION: Dict[str, str] = {
    "SOLVE_SUCCEEDED": "LOCALLY_SOLVED", "SOLVED_TO_ACCEPTABLE_LEVEL": "ALMOST_LOCALLY_SOLVED",
    "SEARCH_DIRECTION_BECOMES_TOO_SMALL": "SLOW_PROGRESS", "DIVERGING_ITERATES": "INFEASIBLE_OR_UNBOUNDED",
    "INFEASIBLE_PROBLEM_DETECTED": "LOCALLY_INFEASIBLE", "MAXIMUM_ITERATIONS_EXCEEDED": "ITERATION_LIMIT",
    "MAXIMUM_WALLTIME_EXCEEDED": "TIME_LIMIT", "INITIAL": "OPTIMIZE_NOT_CALLED", "RESTORATION_FAILED": "NUMERICAL_ERROR",
    "INVALID_NUMBER_DETECTED": "INVALID_MODEL", "ERROR_IN_STEP_COMPUTATION": "NUMERICAL_ERROR",
    "NOT_ENOUGH_DEGREES_OF_FREEDOM": "INVALID_MODEL", "USER_REQUESTED_STOP": "INTERRUPTED", "INTERNAL_ERROR": "OTHER_ERROR",
    "INVALID_NUMBER_OBJECTIVE": "INVALID_MODEL", "INVALID_NUMBER_GRADIENT": "INVALID_MODEL",
    "INVALID_NUMBER_CONSTRAINTS": "INVALID_MODEL", "INVALID_NUMBER_JACOBIAN": "INVALID_MODEL",
    "INVALID_NUMBER_HESSIAN_LAGRANGIAN": "INVALID_MODEL",
}
MADNLP_RESULT: Dict[str, str] = {"SOLVE_SUCCEEDED": "FEASIBLE_POINT", "SOLVED_TO_ACCEPTABLE_LEVEL": "NEARLY_FEASIBLE_POINT",
                                 "INFEASIBLE_PROBLEM_DETECTED": "INFEASIBLE_POINT"}


def translate_termination_status(solver, status) -> str:
    """``translate_termination_status(solver, status)`` (``:384-387``); a solver may bring its own table as
    ``solver.termination_statuses`` (what the MadNLP extension does by dispatch)."""
    table = getattr(solver, "termination_statuses", None) or TERMINATION
    return table.get(str(status), "OTHER_ERROR")


def translate_result_status(solver, status) -> str:
    table = getattr(solver, "result_statuses", None) or RESULT
    return table.get(str(status), "UNKNOWN_RESULT_STATUS")


# per solver family: (print-level option, its default, its silent value, wall-time option, its default)
OPTION_CONVENTIONS = {
    "ipopt": ("print_level", 5, 0, "max_wall_time", 1.0e20),          # ext/InfiniteExaModelsIpopt.jl:5-7
    "madnlp": ("print_level", "INFO", "ERROR", "max_wall_time", 1.0e6),   # ext/InfiniteExaModelsMadNLP.jl:6-8
}


def process_options(options: Dict[str, Any], backend, convention=OPTION_CONVENTIONS["ipopt"]) -> Dict[str, Any]:
    """``_process_options(options, backend)`` of the solver extensions: only options that are new or changed since the
    previous solve are handed on; ``silent`` and ``time_limit`` of the backend become the solver's print-level / wall-time
    options, and are restored to the solver's defaults once they are switched off again.  Updates ``backend.prev_options``."""
    pl, pl_default, pl_silent, wt, wt_default = convention
    prev = backend.prev_options
    new = {k: v for k, v in options.items() if k not in prev or prev[k] != v}
    if backend.silent and prev.get(pl, pl_default) != pl_silent:
        new[pl] = pl_silent
    elif not backend.silent and prev.get(pl, pl_default) == pl_silent and pl not in options:
        new[pl] = pl_default
    limit = backend.time_limit
    if not np.isnan(limit) and prev.get(wt, float("nan")) != limit:
        new[wt] = limit
    elif wt not in options and np.isnan(limit) and prev.get(wt, wt_default) != wt_default:
        new[wt] = wt_default
    prev.update(new)
    return new
