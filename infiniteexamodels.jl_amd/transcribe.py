"""InfiniteModel → ExaCore: host mirror of ``/root/reference/src/transform.jl``.

Function names and the build order follow the reference one-to-one
(``build_exa_core!``, ``transform.jl:771-796``) so that template order — hence row
offsets and the COO block order — is the reference's:

  1 ``_build_base_iterators``        (:2-38)
  2 ``_add_finite_parameters``       (:120-131)
  3 ``_add_finite_variables``        (:104-117)
  4 ``_add_infinite_variables``      (:134-158)  variables, then derivative variables
  5 ``_add_parameter_functions``     (:161-183)
  6 ``_add_semi_infinite_variables`` (:235-256)   7 ``_add_point_variables`` (:273-287)
  8 ``_add_constraints``             (:420-462)
  9 ``_add_derivative_approximations`` (:511-562)
 10 ``_add_collocation_restrictions``  (:565-601)  (orthogonal collocation: not yet built)
 11 ``_add_objective``               (:708-767)
"""
from __future__ import annotations

import warnings
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import nodes as N
from .core import ExaCore, Parameter, Variable
from .infinite import (DerivativeRef, FiniteParameterRef, FiniteVariableRef, InfiniteModel,
                       InfiniteParameterRef, InfiniteVariableRef, MeasureRef, ParameterFunctionRef,
                       PointVariableRef, SemiInfiniteVariableRef, VarInfo,
                       parameter_group_int_indices)
from .items import Field, Items
from .jump_expr import (AffExpr, NonlinearExpr, QuadExpr, VariableRef, all_expression_variables,
                        is_number, map_expression)
from .operators import nl_op

_ObjMeasureExpansionWarn = (
    "Unable to convert objective measures into a form that is efficient for ExaModels using "
    "existing heuristics. Performance may be significantly degraded. Try simplying the objective "
    "structure. if you think this form should be supported, please open an issue.")


class ExaMappingData:
    """``InfiniteExaModels.ExaMappingData`` (``src/infiniteopt_backend.jl:12-57``)."""

    def __init__(self):
        self.infvar_mappings: Dict[VariableRef, Variable] = {}
        self.finvar_mappings: Dict[VariableRef, N.Var] = {}
        self.param_mappings: Dict[VariableRef, Parameter] = {}
        self.constraint_mappings: Dict[object, object] = {}
        self.param_alias: Dict[VariableRef, str] = {}
        self.group_alias: List[str] = []
        self.base_itrs: List[Items] = []
        self.support_to_index = _SupportIndex()
        self.semivar_info: Dict[VariableRef, tuple] = {}
        self.has_internal_supps: List[bool] = []
        self.support_labels: List[list] = []
        # (Variable, group indices) in creation order — used by the sharding maps
        self.finvar_slabs: List[tuple] = []
        self.infvar_slabs: List[tuple] = []
        self.obj_terms = 0
        self.model = None


def _supp_key(values) -> tuple:
    return tuple(float(v) for v in np.atleast_1d(values))


class _SupportIndex:
    """``(group index, support tuple) -> 1-based support index`` (``data.support_to_index`` of the
    reference, a Dict filled support by support at transform.jl:24-29).  Looked up lazily: point
    and semi-infinite variables ask for a handful of supports, a 10^6-entry Python dict per group
    costs seconds.  Small groups are turned into a dict on first use, large ones are searched
    with one vectorised comparison per (memoised) query."""

    def __init__(self):
        self._supports: Dict[int, np.ndarray] = {}
        self._dicts: Dict[int, dict] = {}
        self._memo: Dict[tuple, Optional[int]] = {}

    def add_group(self, gindex: int, supports: np.ndarray) -> None:
        self._supports[gindex] = np.asarray(supports, dtype=np.float64).reshape(len(supports), -1)

    def get(self, key, default=None):
        if key in self._memo:
            hit = self._memo[key]
            return default if hit is None else hit
        g, supp = key
        S = self._supports.get(g)
        hit = None
        if S is not None and len(supp) == S.shape[1]:
            if S.shape[0] <= 4096:
                d = self._dicts.get(g)
                if d is None:
                    d = self._dicts[g] = {tuple(float(v) for v in row): i + 1 for i, row in enumerate(S)}
                hit = d.get(tuple(supp))
            else:
                rows = np.flatnonzero((S == np.asarray(supp, dtype=np.float64)).all(axis=1))
                hit = int(rows[-1]) + 1 if rows.size else None      # a Dict keeps the last duplicate
        self._memo[key] = hit
        return default if hit is None else hit

    def __getitem__(self, key):
        hit = self.get(key)
        if hit is None:
            raise KeyError(key)
        return hit

    def __contains__(self, key):
        return self.get(key) is not None


# 1 ---------------------------------------------------------------------------
def _build_base_iterators(data: ExaMappingData, m: InfiniteModel) -> None:
    raw = 0
    for g in m.groups:
        raw += 1
        m.add_generative_supports(g)        # transform.jl:22
        for pref in g.prefs:
            data.param_alias[pref] = f"dp{raw}{pref.pos + 1}" if g.dependent else f"ip{raw}"
        itr_sym = f"group_idx{len(data.group_alias) + 1}"
        data.group_alias.append(itr_sym)
        data.support_to_index.add_group(g.index, g.supports)
        vals = {data.param_alias[p]: np.ascontiguousarray(g.supports[:, p.pos]) for p in g.prefs}
        data.base_itrs.append(Items.from_supports(itr_sym, g.num_supports, vals, group_id=g.index))
        data.has_internal_supps.append(bool(g.internal is not None and g.internal.any()))   # transform.jl:35


# bounds / start ----------------------------------------------------------------
def _eval_over_supports(val, m: InfiniteModel, group_idxs: List[int], dims: Tuple[int, ...]) -> np.ndarray:
    """Evaluate a function-valued bound/start over the support product
    (``_get_variable_bounds_and_start``, transform.jl:78-101)."""
    cols = []
    for ax, gi in enumerate(group_idxs):
        g = m.groups[gi - 1]
        for c in range(g.supports.shape[1]):
            shape = [1] * len(dims)
            shape[ax] = dims[ax]
            cols.append(g.supports[:, c].reshape(shape))
    try:
        out = np.asarray(val(*cols), dtype=np.float64)
        return np.broadcast_to(out, dims).copy()
    except Exception:
        out = np.zeros(dims)
        for idx in np.ndindex(*dims):
            supp = [float(c[tuple(i if c.shape[a] > 1 else 0 for a, i in enumerate(idx))]) for c in cols]
            out[idx] = val(*supp)
        return out


def _get_variable_bounds_and_start(info: VarInfo, m: InfiniteModel = None, group_idxs=None, dims=None):
    lb, ub, start = -np.inf, np.inf, 0.0

    def proc(v):
        return _eval_over_supports(v, m, group_idxs, dims) if callable(v) else v

    if info.fix is not None:
        lb = ub = proc(info.fix)
    if info.lb is not None:
        lb = proc(info.lb)
    if info.ub is not None:
        ub = proc(info.ub)
    if info.start is not None:
        start = proc(info.start)
    return lb, ub, start


# 2-5 -----------------------------------------------------------------------------
def _add_finite_parameters(core: ExaCore, data: ExaMappingData, m: InfiniteModel) -> None:
    for pref in m.finite_parameters:
        data.param_mappings[pref] = core.add_par([pref.value])


def _add_finite_variables(core: ExaCore, data: ExaMappingData, m: InfiniteModel) -> None:
    for vref in m.finite_variables:
        lb, ub, start = _get_variable_bounds_and_start(vref.info)
        new_var = core.add_var(1, start=start, lvar=lb, uvar=ub)
        data.finvar_mappings[vref] = new_var[1]
        data.finvar_slabs.append((new_var, []))


def _add_infinite_variables(core: ExaCore, data: ExaMappingData, m: InfiniteModel) -> None:
    vrefs = list(m.infinite_variables) + list(m.derivatives)
    # one allocation for all slabs (the sizes are known before the first add_var)
    core.reserve_vars(sum(int(np.prod([len(data.base_itrs[g - 1]) for g in v.group_idxs], dtype=np.int64)) for v in vrefs))
    for vref in vrefs:
        group_idxs = vref.group_idxs
        dims = tuple(len(data.base_itrs[g - 1]) for g in group_idxs)
        lb, ub, start = _get_variable_bounds_and_start(vref.info, m, group_idxs, dims)
        data.infvar_mappings[vref] = core.add_var(*dims, start=start, lvar=lb, uvar=ub, groups=group_idxs)
        data.infvar_slabs.append((data.infvar_mappings[vref], list(group_idxs)))


def _add_parameter_functions(core: ExaCore, data: ExaMappingData, m: InfiniteModel) -> None:
    for pfref in m.parameter_functions:
        group_idxs = pfref.group_idxs
        dims = tuple(len(data.base_itrs[g - 1]) for g in group_idxs)
        vals = _eval_over_supports(pfref.func, m, group_idxs, dims)
        data.param_mappings[pfref] = core.add_par(vals)


# 6-7 -----------------------------------------------------------------------------
def _process_semi_infinite_var(vref: SemiInfiniteVariableRef, data: ExaMappingData):
    ivref = vref.ivref
    orig_groups = ivref.group_idxs
    free_groups = vref.group_idxs
    indexing: List[object] = []
    for g in orig_groups:
        if g in free_groups:
            indexing.append(data.group_alias[g - 1])
        else:
            supp = [a for a, p in zip(vref.args, ivref.prefs) if p.group.index == g]
            indexing.append(data.support_to_index[(g, _supp_key(supp))])
    mapped = data.param_mappings[ivref] if isinstance(ivref, ParameterFunctionRef) else data.infvar_mappings[ivref]
    data.semivar_info[vref] = (mapped, indexing)
    return data.semivar_info[vref]


def _update_bounds_and_start(core: ExaCore, info: VarInfo, var: N.Var) -> None:
    i = int(var.i) - 1
    if info.lb is not None:
        core.lvar[i] = -np.inf if np.isnan(info.lb) else info.lb
    if info.ub is not None:
        core.uvar[i] = np.inf if np.isnan(info.ub) else info.ub
    if info.fix is not None:
        core.lvar[i] = core.uvar[i] = info.fix
    if info.start is not None:
        core.x0[i] = info.start


def _add_semi_infinite_variables(core: ExaCore, data: ExaMappingData, m: InfiniteModel) -> None:
    for vref in m.semi_infinite_variables:
        mapped, indexing = _process_semi_infinite_var(vref, data)
        info = vref.info
        if any(v is not None for v in (info.lb, info.ub, info.fix, info.start)):
            ranges = [[idx] if isinstance(idx, int) else range(1, mapped.size[i] + 1)
                      for i, idx in enumerate(indexing)]
            for idx in np.ndindex(*[len(r) for r in ranges]):
                var = mapped[tuple(r[j] for r, j in zip(ranges, idx))]
                _update_bounds_and_start(core, info, var)


def _process_point_var(vref: PointVariableRef, data: ExaMappingData) -> N.Var:
    ivref = vref.ivref
    idxs = []
    for g in ivref.group_idxs:
        supp = [a for a, p in zip(vref.values, ivref.prefs) if p.group.index == g]
        idxs.append(data.support_to_index[(g, _supp_key(supp))])
    return data.infvar_mappings[ivref][tuple(idxs)]


def _add_point_variables(core: ExaCore, data: ExaMappingData, m: InfiniteModel) -> None:
    for vref in m.point_variables:
        try:
            pt = _process_point_var(vref, data)
        except KeyError:
            if _shard(m) is not None:
                continue   # the point lies outside this rank's support window
            raise
        data.finvar_mappings[vref] = pt
        _update_bounds_and_start(core, vref.info, pt)


# _map_variable / _exafy ------------------------------------------------------------
def _map_variable(vref: VariableRef, data_src, data: ExaMappingData):
    if isinstance(vref, FiniteVariableRef):
        return data.finvar_mappings[vref]
    if isinstance(vref, PointVariableRef):
        if vref not in data.finvar_mappings:
            data.finvar_mappings[vref] = _process_point_var(vref, data)
        return data.finvar_mappings[vref]
    if isinstance(vref, InfiniteVariableRef):      # infinite variable or derivative
        idx = tuple(data_src[data.group_alias[g - 1]] for g in vref.group_idxs)
        return data.infvar_mappings[vref][idx]
    if isinstance(vref, SemiInfiniteVariableRef):
        if vref not in data.semivar_info:
            _process_semi_infinite_var(vref, data)
        ivar, inds = data.semivar_info[vref]
        return ivar[tuple(i if isinstance(i, int) else data_src[i] for i in inds)]
    if isinstance(vref, InfiniteParameterRef):
        return data_src[data.param_alias[vref]]
    if isinstance(vref, FiniteParameterRef):
        return data.param_mappings[vref][1]
    if isinstance(vref, ParameterFunctionRef):
        idx = tuple(data_src[data.group_alias[g - 1]] for g in vref.group_idxs)
        return data.param_mappings[vref][idx]
    raise TypeError(f"Unable to add `{vref!r}` to an ExaModel, it's index type `{vref.kind}` "
                    "is not yet supported by InfiniteExaModels.")


def _sum(gen):
    """Julia ``sum(generator)``: left fold without an initial zero."""
    acc = None
    for v in gen:
        acc = v if acc is None else acc + v
    return acc


def _exafy(expr, data_src, data: ExaMappingData):
    """``_exafy`` (transform.jl:337-389)."""
    if isinstance(expr, VariableRef):
        if isinstance(expr, MeasureRef):
            raise TypeError("measures must be expanded before `_exafy`")
        return _map_variable(expr, data_src, data)
    if is_number(expr):
        return expr
    if isinstance(expr, AffExpr):
        c = expr.constant
        if expr.terms:
            ex = _sum((_exafy(v, data_src, data) if coef == 1.0 else coef * _exafy(v, data_src, data))
                      for coef, v in expr.linear_terms())
            return ex if c == 0.0 else ex + c
        return c
    if isinstance(expr, QuadExpr):
        aff = _exafy(expr.aff, data_src, data)
        if expr.terms:
            def term(c, v1, v2):
                if v1 is v2:
                    v_ex = _exafy(v1, data_src, data)
                    return N.FUNCS["abs2"](v_ex) if c == 1.0 else c * N.FUNCS["abs2"](v_ex)
                v1_ex = _exafy(v1, data_src, data)
                v2_ex = _exafy(v2, data_src, data)
                return v1_ex * v2_ex if c == 1.0 else c * v1_ex * v2_ex
            ex = _sum(term(c, v1, v2) for c, v1, v2 in expr.quad_terms())
            return ex if expr.aff.is_zero() else ex + aff
        return aff
    if isinstance(expr, NonlinearExpr):
        return nl_op(expr.head)(*[_exafy(a, data_src, data) for a in expr.args])
    raise TypeError(f"cannot transcribe expression of type {type(expr).__name__}")


def _finalize_expr(expr):
    return N.Null(expr) if is_number(expr) else expr


# expand_measures fallback ----------------------------------------------------------------
def _reduce_ref(v, fixed: Dict[VariableRef, float]):
    """Restrict one reference to parameter values ``fixed`` (pref → value): what InfiniteOpt's
    measure expansion does with ``make_point_variable_ref`` / ``make_semi_infinite_variable_ref``."""
    if isinstance(v, InfiniteParameterRef):
        return fixed.get(v, v)
    if isinstance(v, (InfiniteVariableRef, ParameterFunctionRef)):
        if not any(p in fixed for p in v.prefs):
            return v
        args = [fixed.get(p, p) for p in v.prefs]
        if isinstance(v, ParameterFunctionRef) and all(is_number(a) for a in args):
            return float(v.func(*args))
        return v(*args)
    if isinstance(v, SemiInfiniteVariableRef):
        if not any((not is_number(a)) and a in fixed for a in v.args):
            return v
        args = [a if is_number(a) else fixed.get(a, a) for a in v.args]
        if isinstance(v.ivref, ParameterFunctionRef) and all(is_number(a) for a in args):
            return float(v.ivref.func(*args))
        return v.ivref(*args)
    return v


def expand_measures(expr, m: InfiniteModel):
    """``InfiniteOpt.expand_measures`` [EXT]: every measure becomes the explicit weighted sum of
    its integrand at the measure's supports (the reference's slow fallback, transform.jl:435,
    :719, :758)."""
    def expand_one(mref: MeasureRef):
        supps, coeffs = _measure_data(mref)
        inner = expand_measures(mref.func, m)
        total = None
        for k in range(len(coeffs)):
            fixed = {p: float(supps[k, c]) for c, p in enumerate(mref.prefs)}
            term = map_expression(lambda v: _reduce_ref(v, fixed), inner) if not is_number(inner) else inner
            term = coeffs[k] * term
            total = term if total is None else total + term
        return total if total is not None else 0.0

    if is_number(expr):
        return expr
    return map_expression(lambda v: expand_one(v) if isinstance(v, MeasureRef) else v, expr)


# sharding hooks (shard.py) -----------------------------------------------------------
def _shard(m: InfiniteModel):
    return getattr(m, "shard", None)


def owned(g: int, data: ExaMappingData, m: InfiniteModel) -> Items:
    """Base iterator of group ``g`` cut to the supports this rank owns."""
    sp = _shard(m)
    itr = data.base_itrs[g - 1]
    if sp is not None and g == sp.group_index:
        return itr.select(sp.own_lo, sp.own_n)
    return itr


def _rank0_only(group_idxs, m: InfiniteModel) -> bool:
    """Templates that do not involve the sharded parameter live on rank 0 only."""
    sp = _shard(m)
    return sp is not None and sp.group_index not in group_idxs and sp.rank != 0


def _emit_here(expr, group_idxs, data: ExaMappingData, m: InfiniteModel) -> bool:
    """Does this rank own the template?  Templates over the sharded parameter: every rank
    (cut to its supports).  Templates that do not iterate over it: the rank that owns the
    supports of their point variables (``x(0) == 0`` → the rank holding t = 0), else rank 0."""
    sp = _shard(m)
    if sp is None or sp.group_index in group_idxs:
        return True
    owners = set()
    for v in all_expression_variables(expr):
        if isinstance(v, PointVariableRef) and sp.group_index in v.ivref.group_idxs:
            supp = [a for a, p in zip(v.values, v.ivref.prefs) if p.group.index == sp.group_index]
            i = data.support_to_index.get((sp.group_index, _supp_key(supp)))
            owners.add(i is not None and sp.own_lo <= i - 1 < sp.own_lo + sp.own_n)
    if owners:
        if len(owners) > 1:
            raise NotImplementedError("a finite template couples point variables owned by different ranks")
        return owners.pop()
    return sp.rank == 0


# 8 ---------------------------------------------------------------------------------
def _product_itr(itrs: List[Items]) -> Items:
    out = itrs[0]
    for it in itrs[1:]:
        out = out.product(it)
    return out


def _restriction_mask(restriction, itr: Items, data: ExaMappingData) -> np.ndarray:
    cols = [itr.column(data.param_alias[p]) for p in restriction.parameter_refs]
    try:
        mask = np.asarray(restriction.func(*cols))
        if mask.shape == cols[0].shape and mask.dtype == bool:
            return mask
    except Exception:
        pass
    return np.array([bool(restriction.func(*[float(c[k]) for c in cols])) for k in range(len(itr))])


def id_index(lst, obj) -> int:
    for i, o in enumerate(lst):
        if o is obj:
            return i
    raise ValueError("object not in list")


def _add_constraints(core: ExaCore, data: ExaMappingData, m: InfiniteModel) -> None:
    for cref in m.constraints:
        expr = cref.func
        if any(isinstance(v, MeasureRef) for v in all_expression_variables(expr)):
            warnings.warn("Constrained measures can lead to poor performance with ExaModels.")
            expr = expand_measures(expr, m)
        group_idxs = parameter_group_int_indices(expr)
        if not _emit_here(expr, group_idxs, data, m):
            continue
        if not group_idxs:
            itr = Items.single()
        elif len(group_idxs) == 1:
            itr = owned(group_idxs[0], data, m)
        else:
            itr = _product_itr([owned(g, data, m) for g in group_idxs])
        if cref.restriction is not None:
            itr = itr.filter(_restriction_mask(cref.restriction, itr, data))
        data_src = N.DataSource()
        em_expr = _finalize_expr(_exafy(expr, data_src, data))
        cref.mapping = core.add_con(em_expr, itr, lcon=cref.lb, ucon=cref.ub)
        core.templates[-1].tag = ("con", id_index(m.constraints, cref))
        data.constraint_mappings[cref] = cref.mapping


# 9 ---------------------------------------------------------------------------------
def _make_reduced_expr(vref, pref, idx, data_src, data: ExaMappingData):
    """``InfiniteOpt.make_reduced_expr`` extension (transform.jl:471-508): the variable
    at support index expression ``idx`` along ``pref``'s group."""
    alias = data.group_alias[pref.group.index - 1]
    if isinstance(vref, SemiInfiniteVariableRef):
        ivar, inds = data.semivar_info[vref]
        return ivar[tuple(i if isinstance(i, int) else (idx if i == alias else data_src[i]) for i in inds)]
    pars = tuple(idx if data.group_alias[g - 1] == alias else data_src[data.group_alias[g - 1]]
                 for g in vref.group_idxs)
    return data.infvar_mappings[vref][pars]


def derivative_expr_data(method: tuple, supps: np.ndarray):
    """``InfiniteOpt.derivative_expr_data`` for finite differences [EXT]:
    (0-based item → support positions, per-item coefficient columns)."""
    n = len(supps)
    kind = method[0]
    if kind == "fd_backward":
        return np.arange(1, n), [supps[1:] - supps[:-1]]
    if kind == "fd_forward":
        return np.arange(0, n - 1), [supps[1:] - supps[:-1]]
    if kind == "fd_central":
        return np.arange(1, n - 1), [supps[2:] - supps[:-2]]
    raise NotImplementedError(f"derivative method {method!r}")


def collocation_items(pref, base_itr: Items, supps: np.ndarray, internal: np.ndarray, data: ExaMappingData) -> Items:
    """Item iterator of an orthogonal-collocation derivative [EXT — InfiniteOpt
    ``evaluate_derivative(::OrthogonalCollocation{GaussLobatto})`` restated]: for every element
    with nodes ``t_0 (lower boundary), t_1..t_n`` one row per node ``j = 1..n``:

        Σ_k Minvᵀ[k, j]·∂y(t_k) − y(t_j) + y(t_0) = 0,   M1ᵀ[k,j] = k·τ_j^{k−1}, M2ᵀ[k,j] = τ_j^k,
        Minvᵀ = M1ᵀ \\ M2ᵀ,  τ_j = t_j − t_0.

    Items form the box (j fastest, element): integer fields = node index (the group alias),
    ``d_lb`` (lower-boundary index), ``d_n1..d_nn`` (the element's node indices); float fields
    ``d_arg1..d_argn`` = column j of Minvᵀ."""
    cache = data.__dict__.setdefault("_collocation_items", {})
    if pref in cache:                                       # every derivative w.r.t. `pref` shares the iterator
        return cache[pref]
    n_tot = len(supps)
    lb_idx = np.nonzero(~internal)[0][:-1]                 # 0-based lower boundaries
    n = int(lb_idx[1] - lb_idx[0]) if len(lb_idx) > 1 else n_tot - 1
    assert np.all(np.diff(lb_idx) == n) and lb_idx[-1] + n == n_tot - 1
    ne = len(lb_idx)
    # all elements at once (batched LAPACK solve): [element, k, j]
    tau = supps[lb_idx[:, None] + 1 + np.arange(n)[None, :]] - supps[lb_idx][:, None]
    kk = np.arange(1, n + 1)[None, :, None]
    M1t = kk * tau[:, None, :] ** (kk - 1)
    M2t = tau[:, None, :] ** kk
    coef = np.ascontiguousarray(np.linalg.solve(M1t, M2t).transpose(0, 2, 1))   # [element, j, k]
    alias = data.group_alias[pref.group.index - 1]
    p_alias = data.param_alias[pref]
    fields = {
        alias: Field("int", "affine", 2, (1, n)),           # node index (1-based)
        "d_lb": Field("int", "affine", 1, (0, n)),
        p_alias: Field("float", "gather", 1, (1, n), np.ascontiguousarray(supps)),
    }
    for k in range(n):
        fields[f"d_n{k + 1}"] = Field("int", "affine", 2 + k, (0, n))
        fields[f"d_arg{k + 1}"] = Field("float", "gather", 0, (1, n), np.ascontiguousarray(coef[:, :, k].reshape(-1)))
    # a virtual grid shared by every collocation template of this parameter → one fused kernel
    gi = pref.group.index
    cache[pref] = Items((n, ne), fields, grid=((1000 + gi, 2000 + gi), (0, 0)))
    return cache[pref]


def make_indexed_derivative_expr(dref, vref, pref, idx, data_src, data, method: tuple, d_args):
    """``InfiniteOpt.make_indexed_derivative_expr`` [EXT]; first order, finite differences.
    Backward: ``h·∂y[i] − y[i] + y[i−1]`` with ``h = tᵢ − tᵢ₋₁``."""
    d = _make_reduced_expr(dref, pref, idx, data_src, data)
    kind = method[0]
    if kind == "fd_backward":
        return d_args[0] * d - _make_reduced_expr(vref, pref, idx, data_src, data) \
            + _make_reduced_expr(vref, pref, idx - 1, data_src, data)
    if kind == "fd_forward":
        return d_args[0] * d - _make_reduced_expr(vref, pref, idx + 1, data_src, data) \
            + _make_reduced_expr(vref, pref, idx, data_src, data)
    if kind == "fd_central":
        return d_args[0] * d - _make_reduced_expr(vref, pref, idx + 1, data_src, data) \
            + _make_reduced_expr(vref, pref, idx - 1, data_src, data)
    if kind == "oc":
        n = len(d_args)
        acc = None
        for k in range(n):
            term = d_args[k] * _make_reduced_expr(dref, pref, data_src[f"d_n{k + 1}"], data_src, data)
            acc = term if acc is None else acc + term
        return acc - _make_reduced_expr(vref, pref, idx, data_src, data) \
            + _make_reduced_expr(vref, pref, data_src["d_lb"], data_src, data)
    raise NotImplementedError(kind)


def _add_derivative_approximations(core: ExaCore, data: ExaMappingData, m: InfiniteModel) -> None:
    for dref in m.derivatives:
        vref, pref = dref.arg, dref.pref
        method = pref.group.derivative_method
        group_idxs = vref.group_idxs
        pref_group = pref.group.index
        p_alias = data.param_alias[pref]
        base_itr = data.base_itrs[pref_group - 1]
        supps = base_itr.column(p_alias)
        if pref.group.dependent:
            order = np.argsort(supps, kind="stable")
            srt_itr, supps = base_itr.take(order), supps[order]
        else:
            srt_itr = base_itr
        if method[0] == "oc":
            if pref.group.internal is None:
                raise RuntimeError("collocation supports were not generated")
            pref_itr = collocation_items(pref, srt_itr, supps, pref.group.internal, data)
            aliases = [f"d_arg{i + 1}" for i in range(max(1, method[1] - 1))]
            idxs = np.arange(1, len(supps))
        else:
            idxs, arg_cols = derivative_expr_data(method, supps)
            aliases = [f"d_arg{i + 1}" for i in range(len(arg_cols))]
            pref_itr = srt_itr.take(idxs)
            for a, col in zip(aliases, arg_cols):
                pref_itr = pref_itr.with_float(a, col)
        sp = _shard(m)
        if sp is not None and sp.group_index == pref_group:
            # the window carries the stencil halo, so every local row of a backward difference
            # belongs to an owned support
            if method[0] != "fd_backward":
                raise NotImplementedError("sharding along a parameter supports backward differences only")
            keep = idxs >= sp.own_lo
            assert keep.all(), "halo does not cover the stencil"
        if _rank0_only(group_idxs, m):
            continue
        if len(group_idxs) > 1:
            itr = _product_itr([pref_itr if g == pref_group else owned(g, data, m) for g in group_idxs])
        else:
            itr = pref_itr
        data_src = N.DataSource()
        em_expr = make_indexed_derivative_expr(
            dref, vref, pref, data_src[data.group_alias[pref_group - 1]], data_src, data, method,
            [data_src[a] for a in aliases])
        core.add_con(em_expr, itr)
        core.templates[-1].tag = ("deriv", id_index(m.derivatives, dref))


# 10 --------------------------------------------------------------------------------
def _add_collocation_restrictions(core: ExaCore, data: ExaMappingData, m: InfiniteModel) -> None:
    """Piecewise-constant variables over collocation elements (transform.jl:565-601):
    ``v[i1] − v[i2] == 0`` with ``i1`` the element's upper boundary, ``i2`` each internal node."""
    for pref, vrefs in m.piecewise_vars.items():
        g = pref.group
        if g.internal is None or not g.internal.any():
            continue                                                           # :573-575
        pref_group = g.index
        pref_alias = data.group_alias[pref_group - 1]
        num_nodes = int(g.derivative_method[1]) - 2                              # internal nodes per element (:578)
        num_supps = g.num_supports
        ubs = np.repeat(np.arange(2 + num_nodes, num_supps + 1, num_nodes + 1), num_nodes)   # :580
        ub_set = set(ubs.tolist())
        pts = np.array([i for i in range(2, num_supps) if i not in ub_set])                 # :581
        assert len(ubs) == len(pts)
        recs = Items.from_records([dict(i1=int(a), i2=int(b)) for a, b in zip(ubs, pts)])   # :582
        recs.grid = ((3000 + pref_group,), (0,))   # virtual grid: the restrictions of all variables fuse
        for k, vref in enumerate(vrefs):
            group_idxs = vref.group_idxs
            itrs = [recs if gi == pref_group else data.base_itrs[gi - 1] for gi in group_idxs]
            itr = _product_itr(itrs) if len(itrs) > 1 else itrs[0]
            data_src = N.DataSource()
            ivar = data.infvar_mappings[vref]
            idx1 = tuple(data_src["i1"] if data.group_alias[gi - 1] == pref_alias else data_src[data.group_alias[gi - 1]]
                         for gi in group_idxs)
            idx2 = tuple(data_src["i2"] if data.group_alias[gi - 1] == pref_alias else data_src[data.group_alias[gi - 1]]
                         for gi in group_idxs)
            core.add_con(ivar[idx1] - ivar[idx2], itr)                                       # :593-597
            core.templates[-1].tag = ("colloc", id_index(list(m.piecewise_vars), pref), k)


# 11 --------------------------------------------------------------------------------
def _add_generic_objective_term(core: ExaCore, expr, data: ExaMappingData):
    data.obj_terms += 1
    sp = _shard(data.model) if data.model is not None else None
    if sp is not None and sp.rank != 0:
        return None   # finite objective terms are counted once (rank 0)
    em_expr = _finalize_expr(_exafy(expr, {}, data))
    out = core.add_obj(em_expr, Items.single())
    core.templates[-1].tag = ("obj", data.obj_terms)
    return out


def _measure_data(mref: MeasureRef):
    """supports (n, len(prefs)) and coefficients of a measure [EXT: InfiniteOpt
    UniTrapezoid / sample-average defaults]."""
    g = mref.prefs[0].group
    if mref.method == "trapezoid":
        s = g.supports[:, 0]
        d = np.diff(s)
        c = np.zeros_like(s)
        c[:-1] += d / 2
        c[1:] += d / 2
        return s.reshape(-1, 1), c
    n = g.num_supports
    return g.supports[:, [p.pos for p in mref.prefs]], np.full(n, 1.0 / n)


def _make_measure_itr(mref: MeasureRef, data: ExaMappingData) -> Items:
    supps, coeffs = _measure_data(mref)
    g = mref.prefs[0].group
    assert len(mref.prefs) == len(g.prefs)  # no partially measured dependent parameters (:628)
    itr = data.base_itrs[g.index - 1]      # measure supports == group supports, in order
    sp = _shard(data.model) if data.model is not None else None
    if sp is not None and sp.group_index == g.index:
        if sp.coeffs is not None:
            coeffs = sp.coeffs             # weights of the local supports on the GLOBAL grid
        return itr.with_float("c", coeffs).select(sp.own_lo, sp.own_n)
    return itr.with_float("c", coeffs)


def _has_variable(expr, vref) -> bool:
    return any(v is vref for v in all_expression_variables(expr))


def _terms_can_be_moved_inside_measure(expr, mref) -> bool:
    if isinstance(expr, (VariableRef, AffExpr)):
        return True
    if isinstance(expr, QuadExpr):
        return not any((k.a is mref and k.b is mref) for k in expr.terms)
    if isinstance(expr, NonlinearExpr):
        m_inds = [a for a in expr.args if not is_number(a) and _has_variable(a, mref)]
        if expr.head in ("+", "-"):
            return all(_terms_can_be_moved_inside_measure(a, mref) for a in m_inds)
        if expr.head == "*":
            return len(m_inds) <= 1 and _terms_can_be_moved_inside_measure(m_inds[0], mref)
        return False
    return False


def _process_measure_sum(vref: MeasureRef, data: ExaMappingData, prev_itr: Optional[Items] = None):
    mexpr = vref.func
    curr_itr = _make_measure_itr(vref, data)
    if prev_itr is None:
        itr = curr_itr
    else:
        # [(i[1]..., i[2]..., c = i[1].c * i[2].c) for i in product(curr_itr, prev_itr)]
        itr = curr_itr.product(prev_itr)
        cc = np.outer(prev_itr.column("c"), curr_itr.column("c")).reshape(-1)
        steps, s = [], 1
        for n in itr.dims:
            steps.append(s)
            s *= n
        itr.fields["c"] = Field("float", "gather", 0, tuple(steps), np.ascontiguousarray(cc))
    vrefs = all_expression_variables(mexpr)
    mrefs = [v for v in vrefs if isinstance(v, MeasureRef)]
    if not mrefs:
        return mexpr, itr
    if len(mrefs) == 1 and _terms_can_be_moved_inside_measure(mexpr, mrefs[0]):
        mref = mrefs[0]
        inner_mexpr, new_itr = _process_measure_sum(mref, data, itr)
        return map_expression(lambda v: inner_mexpr if v is mref else v, mexpr), new_itr
    warnings.warn(_ObjMeasureExpansionWarn)
    return expand_measures(mexpr, data.model), itr


def _add_objective_aff_term(core: ExaCore, coef, vref: VariableRef, data: ExaMappingData) -> None:
    if isinstance(vref, MeasureRef):
        mexpr, itr = _process_measure_sum(vref, data)
        data.obj_terms += 1
        sp = _shard(data.model) if data.model is not None else None
        if sp is not None and sp.rank != 0 and f"group_idx{sp.group_index}" not in itr.fields:
            return   # a measure that does not run over the sharded parameter: rank 0 only
        data_src = N.DataSource()
        em_expr = data_src.c * _exafy(coef * mexpr, data_src, data)
        core.add_obj(_finalize_expr(em_expr), itr)
        core.templates[-1].tag = ("obj", data.obj_terms)
    else:
        _add_generic_objective_term(core, coef * vref, data)


def _add_objective(core: ExaCore, expr, data: ExaMappingData, m: InfiniteModel) -> None:
    if isinstance(expr, VariableRef):
        _add_objective_aff_term(core, 1.0, expr, data)
    elif isinstance(expr, AffExpr):
        for coef, vref in expr.linear_terms():
            _add_objective_aff_term(core, coef, vref, data)
        if expr.constant != 0.0:
            data.obj_terms += 1
            sp = _shard(m)
            if sp is None or sp.rank == 0:
                core.add_obj(N.Null(expr.constant))
                core.templates[-1].tag = ("obj", data.obj_terms)
    elif isinstance(expr, QuadExpr):
        for coef, v1, v2 in expr.quad_terms():
            if isinstance(v1, MeasureRef) and isinstance(v2, MeasureRef):
                warnings.warn(_ObjMeasureExpansionWarn)
                _add_generic_objective_term(core, expand_measures(coef * v1 * v2, m), data)
            elif isinstance(v1, MeasureRef):
                _add_objective_aff_term(core, coef * v2, v1, data)
            else:
                _add_objective_aff_term(core, coef * v1, v2, data)
        _add_objective(core, expr.aff, data, m)
    else:
        if any(isinstance(v, MeasureRef) for v in all_expression_variables(expr)):
            warnings.warn(_ObjMeasureExpansionWarn)
        _add_generic_objective_term(core, expand_measures(expr, m), data)


def build_exa_core(core: ExaCore, data: ExaMappingData, m: InfiniteModel) -> ExaCore:
    """``build_exa_core!`` (transform.jl:771-796)."""
    data.model = m
    _build_base_iterators(data, m)
    _add_finite_parameters(core, data, m)
    _add_finite_variables(core, data, m)
    _add_infinite_variables(core, data, m)
    _add_parameter_functions(core, data, m)
    _add_semi_infinite_variables(core, data, m)
    _add_point_variables(core, data, m)
    _add_constraints(core, data, m)
    _add_derivative_approximations(core, data, m)
    _add_collocation_restrictions(core, data, m)
    if m.objective_sense is not None:
        _add_objective(core, m.objective_function, data, m)
    return core


def exa_core(m: InfiniteModel, data: Optional[ExaMappingData] = None, backend=None) -> ExaCore:
    """``ExaModels.ExaCore(inf_model, data; backend)`` (transform.jl:808-817)."""
    data = data if data is not None else ExaMappingData()
    core = ExaCore(backend=backend, minimize=(m.objective_sense != "max"))
    return build_exa_core(core, data, m)
