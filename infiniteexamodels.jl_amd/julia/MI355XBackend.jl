# MI355XBackend.jl — the Julia side of the drop-in boundary (UNEXECUTED: there is no
# `julia` in the build image; written against include/iem.h and the reference's plug point).
#
# Usage in the reference (README.md:41 has `backend = CUDABackend()` in this slot):
#
#     using InfiniteOpt, InfiniteExaModels, MadNLP, AMDGPU
#     include("MI355XBackend.jl"); using .MI355X
#     model = InfiniteModel(ExaTranscriptionBackend(MadNLPSolver; backend = MI355XBackend()))
#
# `ExaTranscriptionBackend` stores the value untyped (`src/infiniteopt_backend.jl:100`) and
# forwards it to `ExaModels.ExaCore(...; backend)` (`src/transform.jl:815`); the two methods
# below intercept `ExaCore`/`ExaModel` construction for `MI355XBackend`, so the transcriber
# (`build_exa_core!`, `src/transform.jl:771-796`) runs unchanged on a host core and the
# finished core is handed to libiem_hip.so as a blob.
module MI355X

using ExaModels, NLPModels, AMDGPU

export MI355XBackend, MI355XModel

const LIBIEM = get(ENV, "LIBIEM_HIP", "libiem_hip.so")

struct MI355XBackend
    device::Int
end
MI355XBackend() = MI355XBackend(0)

struct IemError <: Exception
    code::Cint
    msg::String
end
function check(rc::Cint)
    rc == 0 && return
    throw(IemError(rc, unsafe_string(ccall((:iem_last_error, LIBIEM), Cstring, ()))))
end

# struct iem_meta_t (include/iem.h)
struct IemMeta
    nvar::Int64; ncon::Int64; npar::Int64; nnzj::Int64; nnzh::Int64; n_templates::Int64
    minimize::Int32; n_kernels::Int32
end

mutable struct MI355XModel <: NLPModels.AbstractNLPModel{Float64, ROCVector{Float64}}
    handle::Ptr{Cvoid}
    meta::NLPModels.NLPModelMeta{Float64, ROCVector{Float64}}
    counters::NLPModels.Counters
    θ::Vector{Float64}              # host mirror (infiniteopt_backend.jl:479 reads model.θ)
    core::Any                       # the host ExaCore the transcriber filled
end

host_array(h, which, n) = (out = Vector{Float64}(undef, n);
    check(ccall((:iem_get_host, LIBIEM), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), h, which, out)); out)

# ExaModels.ExaModel(core) at src/infiniteopt_backend.jl:156, for cores built with MI355XBackend
function MI355XModel(core, backend::MI355XBackend)
    blob = to_blob(core)                              # include/iem_blob.h
    href = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:iem_create, LIBIEM), Cint, (Ptr{UInt8}, Csize_t, Cint, Ptr{Ptr{Cvoid}}),
                blob, length(blob), backend.device, href))
    h = href[]
    m = Ref{IemMeta}()
    check(ccall((:iem_meta, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{IemMeta}), h, m))
    mt = m[]
    dev(v) = ROCVector{Float64}(v)
    meta = NLPModels.NLPModelMeta(mt.nvar;
        ncon = mt.ncon, nnzj = mt.nnzj, nnzh = mt.nnzh, minimize = mt.minimize != 0,
        x0 = dev(host_array(h, 0, mt.nvar)), lvar = dev(host_array(h, 1, mt.nvar)),
        uvar = dev(host_array(h, 2, mt.nvar)), lcon = dev(host_array(h, 3, mt.ncon)),
        ucon = dev(host_array(h, 4, mt.ncon)), y0 = dev(host_array(h, 5, mt.ncon)))
    model = MI355XModel(h, meta, NLPModels.Counters(), host_array(h, 6, mt.npar), core)
    finalizer(x -> ccall((:iem_destroy, LIBIEM), Cint, (Ptr{Cvoid},), x.handle), model)
    return model
end

dptr(v::ROCVector{Float64}) = Ptr{Float64}(UInt(pointer(v)))

# ---- NLPModels API: one ccall each ------------------------------------------------------
function NLPModels.obj(m::MI355XModel, x::ROCVector{Float64})
    out = Ref{Float64}()
    check(ccall((:iem_obj, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), m.handle, dptr(x), out))
    return out[]
end
NLPModels.grad!(m::MI355XModel, x::ROCVector{Float64}, g::ROCVector{Float64}) =
    (check(ccall((:iem_grad, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), m.handle, dptr(x), dptr(g))); g)
NLPModels.cons_nln!(m::MI355XModel, x::ROCVector{Float64}, c::ROCVector{Float64}) =
    (check(ccall((:iem_cons, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), m.handle, dptr(x), dptr(c))); c)
NLPModels.jac_coord!(m::MI355XModel, x::ROCVector{Float64}, v::ROCVector{Float64}) =
    (check(ccall((:iem_jac_coord, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), m.handle, dptr(x), dptr(v))); v)
function NLPModels.hess_coord!(m::MI355XModel, x::ROCVector{Float64}, y::ROCVector{Float64},
                               v::ROCVector{Float64}; obj_weight = 1.0)
    check(ccall((:iem_hess_coord, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cdouble, Ptr{Float64}),
                m.handle, dptr(x), dptr(y), obj_weight, dptr(v)))
    return v
end
# jac_coord! + hess_coord! in ONE launch (include/iem.h: iem_jac_hess_coord) for a solver that evaluates both at an accepted
# point (MadNLP); identical bytes to the two calls above
function jac_hess_coord!(m::MI355XModel, x::ROCVector{Float64}, y::ROCVector{Float64}, jac::ROCVector{Float64},
                         hess::ROCVector{Float64}; obj_weight = 1.0)
    check(ccall((:iem_jac_hess_coord, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cdouble, Ptr{Float64}, Ptr{Float64}),
                m.handle, dptr(x), dptr(y), obj_weight, dptr(jac), dptr(hess)))
    return jac, hess
end
# One launch per SOLVER PHASE (include/iem.h: iem_eval_trial / iem_eval_accepted / iem_eval_all; identical bytes to the separate
# calls): obj + cons! at a trial point of the line search, grad! + jac_coord! + hess_coord! at the accepted point
# (ext/InfiniteExaModelsMadNLP.jl:49-50,64).  `eval_trial!` returns f; with `defer = true` it returns at once and `obj_end`
# collects the value (a MadNLP callback that needs c first can overlap the scalar's round trip).
function eval_trial!(m::MI355XModel, x::ROCVector{Float64}, c::ROCVector{Float64}; defer = false)
    out = Ref{Float64}()
    check(ccall((:iem_eval_trial, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                m.handle, dptr(x), dptr(c), defer ? C_NULL : out))
    return defer ? nothing : out[]
end
function eval_accepted!(m::MI355XModel, x::ROCVector{Float64}, y::ROCVector{Float64}, g::ROCVector{Float64}, jac::ROCVector{Float64},
                        hess::ROCVector{Float64}; obj_weight = 1.0)
    check(ccall((:iem_eval_accepted, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                m.handle, dptr(x), dptr(y), obj_weight, dptr(g), dptr(jac), dptr(hess)))
    return g, jac, hess
end
function eval_all!(m::MI355XModel, x::ROCVector{Float64}, y::ROCVector{Float64}, c::ROCVector{Float64}, g::ROCVector{Float64},
                   jac::ROCVector{Float64}, hess::ROCVector{Float64}; obj_weight = 1.0)
    out = Ref{Float64}()
    check(ccall((:iem_eval_all, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                m.handle, dptr(x), dptr(y), obj_weight, dptr(c), dptr(g), dptr(jac), dptr(hess), out))
    return out[]
end
# obj in two halves: obj_begin! first, obj_end after the last launch of the evaluation point — the scalar's host round trip
# overlaps grad!, cons!, jac_coord!, hess_coord! (ext/InfiniteExaModelsIpopt.jl:48-49 evaluates all five at one point)
obj_begin!(m::MI355XModel, x::ROCVector{Float64}) =
    (check(ccall((:iem_obj_begin, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}), m.handle, dptr(x))); m)
function obj_end(m::MI355XModel)
    out = Ref{Float64}()
    check(ccall((:iem_obj_end, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}), m.handle, out))
    return out[]
end
# sharded handles: the halo exchange off the critical path (it rides on the next evaluation launch that takes x and cannot
# touch a halo entry; include/iem.h: iem_halo_exchange_async)
halo_exchange_async!(m::MI355XModel, x::ROCVector{Float64}) =
    (check(ccall((:iem_halo_exchange_async, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}), m.handle, dptr(x))); x)
# optional set-up step, once the solver has allocated its COO value buffers: which of the handle's two code objects
# writes THESE buffers faster (include/iem.h: iem_tune); a no-op for handles with one code object
function tune!(m::MI355XModel, x::ROCVector{Float64}, y::ROCVector{Float64}, jac::ROCVector{Float64},
               hess::ROCVector{Float64}; obj_weight = 1.0)
    check(ccall((:iem_tune, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cdouble, Ptr{Float64}, Ptr{Float64}),
                m.handle, dptr(x), dptr(y), obj_weight, dptr(jac), dptr(hess)))
    return m
end
function NLPModels.jac_structure!(m::MI355XModel, rows::AbstractVector{Int}, cols::AbstractVector{Int})
    r, c = Vector{Int64}(undef, m.meta.nnzj), Vector{Int64}(undef, m.meta.nnzj)
    check(ccall((:iem_jac_structure, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Cint), m.handle, r, c, 1))
    copyto!(rows, r); copyto!(cols, c)
    return rows, cols
end
function NLPModels.hess_structure!(m::MI355XModel, rows::AbstractVector{Int}, cols::AbstractVector{Int})
    r, c = Vector{Int64}(undef, m.meta.nnzh), Vector{Int64}(undef, m.meta.nnzh)
    check(ccall((:iem_hess_structure, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Cint), m.handle, r, c, 1))
    copyto!(rows, r); copyto!(cols, c)
    return rows, cols
end

NLPModels.jprod!(m::MI355XModel, x::ROCVector{Float64}, v::ROCVector{Float64}, Jv::ROCVector{Float64}) =
    (check(ccall((:iem_jprod, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), m.handle, dptr(x), dptr(v), dptr(Jv))); Jv)
NLPModels.jtprod!(m::MI355XModel, x::ROCVector{Float64}, v::ROCVector{Float64}, Jtv::ROCVector{Float64}) =
    (check(ccall((:iem_jtprod, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), m.handle, dptr(x), dptr(v), dptr(Jtv))); Jtv)
function NLPModels.hprod!(m::MI355XModel, x::ROCVector{Float64}, y::ROCVector{Float64}, v::ROCVector{Float64},
                          Hv::ROCVector{Float64}; obj_weight = 1.0)
    check(ccall((:iem_hprod, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Ptr{Float64}),
                m.handle, dptr(x), dptr(y), dptr(v), obj_weight, dptr(Hv)))
    return Hv
end

# ExaModels.set_parameter!(core, param, vals)  (src/infiniteopt_backend.jl:522,546)
function set_parameter!(m::MI355XModel, param, vals)
    v = collect(Float64, vec(vals))
    m.θ[param.offset+1:param.offset+param.length] .= v
    check(ccall((:iem_set_parameter, LIBIEM), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}),
                m.handle, param.offset, length(v), v))
end

# ---- multi-GPU: one process per GPU (e.g. under MPI.jl) ----------------------------------------
# UNEXECUTED, like the rest of this file.  Every rank builds the SAME global core; the library cuts the
# rank's support window (include/iem.h: iem_create_sharded) — `slabs` / `template_groups` as for to_blob.
# `allgather(bytes) -> Vector{UInt8}` is the host's transport (MPI.Allgather, ...): 128 bytes per rank.
function MI355XShardedModel(core, backend::MI355XBackend, group::Int, rank::Int, world::Int, allgather;
                            slabs, template_groups = nothing)
    blob = to_blob(core; slabs = slabs, template_groups = template_groups)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:iem_create_sharded, LIBIEM), Cint,
                (Ptr{UInt8}, Csize_t, Cint, Cint, Cint, Cint, Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}),
                blob, length(blob), backend.device, group, rank, world, C_NULL, 0, h))
    mine = Vector{UInt8}(undef, 128)                       # IEM_COMM_HANDLE_BYTES
    check(ccall((:iem_comm_export, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{UInt8}), h[], mine))
    check(ccall((:iem_comm_connect, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{UInt8}), h[], allgather(mine)))
    return h[]                                             # wrap like MI355XModel(core, backend) (meta from iem_meta)
end
# before cons!/jac_coord!/hess_coord!: the stencil neighbours x_k[a_r - 1] from the left rank (transform.jl:535-557)
halo_exchange!(m::MI355XModel, x::ROCVector{Float64}) =
    (check(ccall((:iem_halo_exchange, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}), m.handle, dptr(x))); x)
# after jtprod! on this rank's rows: the transposed exchange (halo-copy entries -> the left rank's owned entries), then
# allreduce_obj_grad! for the entries of replicated variables
halo_fold!(m::MI355XModel, v::ROCVector{Float64}) =
    (check(ccall((:iem_halo_fold, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}), m.handle, dptr(v))); v)
# after obj (device scalar f) / grad!: objective + gradient entries of replicated variables, summed over the ranks
allreduce_obj_grad!(m::MI355XModel, f::ROCVector{Float64}, g::ROCVector{Float64}) =
    (check(ccall((:iem_allreduce_obj_grad, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), m.handle, dptr(f), dptr(g))); (f, g))

# ---- the linear solver behind MadNLP's AbstractLinearSolver slot (where README.md:36-37 picks CUDSSSolver) ----------
# UNEXECUTED, like the rest of this file; [EXT]: MadNLP's linear-solver interface (factorize!, solve!, is_inertia, inertia)
# from memory of MadNLP 0.8.  One object holds the analysis (grouping by support, coupling rows / columns, gather plan from
# the hess_coord! / jac_coord! value buffers) and the device blocks: include/iem.h, iem_kkt_*.  `hess`, `jac`, `sigma` are the
# solver's own value buffers (MadNLP's sparse KKT system keeps them); delta_w / delta_c its current regularisation.
mutable struct ChainKKTSolver <: Any      # <: MadNLP.AbstractLinearSolver{Float64} once MadNLP is loaded next to this file
    k::Ptr{Cvoid}
    hess::ROCVector{Float64}; jac::ROCVector{Float64}; sigma::ROCVector{Float64}
    delta_w::Float64; delta_c::Float64
    inertia::Vector{Int64}
end
function ChainKKTSolver(m::MI355XModel, hess, jac, sigma)
    k = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:iem_kkt_create, LIBIEM), Cint, (Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}), m.handle, 0, k))   # refuses what it cannot hold
    s = ChainKKTSolver(k[], hess, jac, sigma, 0.0, 0.0, zeros(Int64, 3))
    finalizer(s -> ccall((:iem_kkt_destroy, LIBIEM), Cint, (Ptr{Cvoid},), s.k), s)      # before the model's own finalizer
    return s
end
function factorize!(s::ChainKKTSolver)
    check(ccall((:iem_kkt_assemble, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble),
                s.k, dptr(s.hess), dptr(s.jac), dptr(s.sigma), s.delta_w, s.delta_c))
    check(ccall((:iem_kkt_factor, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Int64}), s.k, s.inertia))
    return s
end
solve!(s::ChainKKTSolver, x::ROCVector{Float64}) =       # in place: the right-hand side is read before the solution is written
    (check(ccall((:iem_kkt_solve, LIBIEM), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), s.k, dptr(x), dptr(x))); x)
is_inertia(::ChainKKTSolver) = true
inertia(s::ChainKKTSolver) = (s.inertia[1], s.inertia[3], s.inertia[2])      # (positive, zero, negative)

# ---- blob writer ----------------------------------------------------------------------------
# Serialises a host ExaCore into the wire format of include/iem_blob.h.  [EXT]: the field and
# type names of ExaModels' internal structs (Objective/Constraint linked lists, SIMDFunction,
# Node1/Node2/Var/ParameterNode, the data-source leaf types) are matched by NAME through
# `nameof(typeof(·))`, so that a rename between ExaModels versions (ParSource/ParIndexed in
# 0.9-0.10, DataSource/DataIndexed in 0.11) needs a one-line change in the sets below.
#
# Node mapping (ExaModels graph types → opcodes of iem_blob.h):
#   Var{I}            → IEM_OP_VAR  + affine index expression over Int item fields
#   ParameterNode{I}  → IEM_OP_PAR  + affine index expression
#   data-indexed leaf → IEM_OP_DATA on a Float64 item column
#   Real              → IEM_OP_CONST
#   Node1{F}          → unary opcode of F;  Node2{F} → binary opcode of F
# Items (`itr`, any iterable of NamedTuples / tuples / scalars) become explicit columns; a column
# that is an arithmetic progression is written as an AFFINE field (no payload), anything else as
# a GATHER field.  Templates whose first Int column is the progression b+1, b+2, … are placed on
# fusion grid 1 with origin b (a scheduling hint only: results never depend on it).

const _SRC_NAMES = (:ParSource, :DataSource)
const _IDX_NAMES = (:ParIndexed, :DataIndexed)
const _BINOP = Dict{Any,Int}(+ => 10, - => 11, * => 12, / => 13, ^ => 14)
const _UNOP = Dict{Any,Int}(
    - => 20, + => 21, inv => 22, sqrt => 23, cbrt => 24, abs => 25, abs2 => 26, exp => 27, exp2 => 28,
    log => 29, log2 => 30, log10 => 31, log1p => 32, sin => 33, cos => 34, tan => 35, asin => 36,
    acos => 37, csc => 38, sec => 39, cot => 40, atan => 41, acot => 42, sind => 43, cosd => 44,
    tand => 45, cscd => 46, secd => 47, cotd => 48, atand => 49, acotd => 50, sinh => 51, cosh => 52,
    tanh => 53, csch => 54, sech => 55, coth => 56, atanh => 57, acoth => 58)

_tname(e) = nameof(typeof(e))
_fn_of(e) = typeof(e).parameters[1].instance            # Node1{F,…} / Node2{F,…}: F = typeof(f)
_w(x::Integer) = reinterpret(UInt64, Int64(x))
_w(x::AbstractFloat) = reinterpret(UInt64, Float64(x))

# key of a data leaf: the chain of getindex/getproperty selectors from the source, e.g. (:t,) or (2, 1)
function _leaf_key(e)
    _tname(e) in _SRC_NAMES && return ()
    _tname(e) in _IDX_NAMES || error("to_blob: not a data leaf: $(typeof(e))")
    return (_leaf_key(getfield(e, :inner))..., typeof(e).parameters[2])
end
_select(item, key::Tuple) = isempty(key) ? item : _select(_pick(item, key[1]), Base.tail(key))
_pick(item, k::Symbol) = getproperty(item, k)
_pick(item, k) = getindex(item, k)

mutable struct _Arrays
    recs::Vector{NTuple{5,Any}}        # (kind, n, payload | nothing, a, b)
    seen::Dict{Any,Int}
end
function _add_array!(A::_Arrays, v::AbstractVector)
    v = v isa AbstractVector{<:Integer} ? collect(Int64, v) : collect(Float64, v)
    if eltype(v) == Float64 && !isempty(v) && all(==(v[1]), v) && !isnan(v[1])
        key = (:fill, length(v), v[1])
        return get!(A.seen, key) do
            push!(A.recs, (2, length(v), nothing, _w(v[1]), UInt64(0))); length(A.recs) - 1
        end
    end
    key = (eltype(v), hash(v), length(v))
    return get!(A.seen, key) do
        push!(A.recs, (eltype(v) == Float64 ? 0 : 1, length(v), v, UInt64(0), UInt64(0))); length(A.recs) - 1
    end
end

mutable struct _Tpl
    items::Vector{Any}
    A::_Arrays
    icol::Dict{Any,Int}; fcol::Dict{Any,Int}
    ifields::Vector{Vector{UInt64}}; ffields::Vector{Vector{UInt64}}
    idx::Vector{Vector{UInt64}}
    nodes::Vector{NTuple{4,UInt64}}
    first_ap::Union{Nothing,Int64}      # base of the first unit-step Int column (fusion hint)
end

function _column_field(T::_Tpl, col::Vector, asint::Bool)
    n = length(col)
    if asint
        c = collect(Int64, col)
        step = n > 1 ? c[2] - c[1] : 0
        if all(k -> c[k] == c[1] + step * (k - 1), 1:n)
            step == 1 && T.first_ap === nothing && (T.first_ap = c[1] - 1)
            return UInt64[_w(0), _w(c[1]), _w(step), _w(0), _w(0), _w(-1)]         # AFFINE
        end
        return UInt64[_w(1), _w(0), _w(1), _w(0), _w(0), _w(_add_array!(T.A, c))]    # GATHER, k-th entry
    end
    return UInt64[_w(1), _w(0), _w(1), _w(0), _w(0), _w(_add_array!(T.A, collect(Float64, col)))]
end
function _ifield!(T::_Tpl, key)
    get!(T.icol, key) do
        push!(T.ifields, _column_field(T, [_select(it, key) for it in T.items], true)); length(T.ifields) - 1
    end
end
function _ffield!(T::_Tpl, key)
    get!(T.fcol, key) do
        push!(T.ffields, _column_field(T, [_select(it, key) for it in T.items], false)); length(T.ffields) - 1
    end
end

# affine index algebra: value = c0 + Σ coef·field
function _affine(T::_Tpl, e)
    e isa Integer && return (Int64(e), Dict{Int,Int64}())
    n = _tname(e)
    (n in _IDX_NAMES || n in _SRC_NAMES) && return (Int64(0), Dict(_ifield!(T, _leaf_key(e)) => Int64(1)))
    if n == :Node2
        f = _fn_of(e); a = getfield(e, :inner1); b = getfield(e, :inner2)
        if f === (+) || f === (-)
            (ca, ta), (cb, tb) = _affine(T, a), _affine(T, b); s = f === (+) ? 1 : -1
            for (k, v) in tb; ta[k] = get(ta, k, 0) + s * v; end
            return (ca + s * cb, ta)
        elseif f === (*) && (a isa Integer || b isa Integer)
            k, (c, t) = a isa Integer ? (a, _affine(T, b)) : (b, _affine(T, a))
            return (k * c, Dict(i => k * v for (i, v) in t))
        end
    elseif n == :Node1 && _fn_of(e) === (-)
        c, t = _affine(T, getfield(e, :inner)); return (-c, Dict(i => -v for (i, v) in t))
    end
    error("to_blob: index expression is not affine in the item fields: $(typeof(e))")
end
function _idx!(T::_Tpl, e)
    c0, terms = _affine(T, e)
    filter!(p -> p.second != 0, terms)
    length(terms) <= 3 || error("to_blob: more than IEM_MAX_IDX_TERMS item fields in one index")
    w = UInt64[_w(c0), _w(length(terms))]
    for (fid, coef) in sort!(collect(terms)); push!(w, _w(fid), _w(coef)); end
    append!(w, fill(UInt64(0), 2 * (3 - length(terms))))
    push!(T.idx, w); return length(T.idx) - 1
end

_node!(T, op, a = 0, b = 0, imm = 0.0) = (push!(T.nodes, (_w(op), _w(a), _w(b), _w(Float64(imm)))); length(T.nodes) - 1)
function _walk!(T::_Tpl, e)
    e isa Real && return _node!(T, 0, 0, 0, e)
    n = _tname(e)
    n == :Null && return _node!(T, 0, 0, 0, something(getfield(e, :value), 0.0))
    n == :Var && return _node!(T, 3, _idx!(T, getfield(e, :i)))
    n == :ParameterNode && return _node!(T, 2, _idx!(T, getfield(e, :i)))
    (n in _IDX_NAMES || n in _SRC_NAMES) && return _node!(T, 1, _ffield!(T, _leaf_key(e)))
    if n == :Node1
        a = _walk!(T, getfield(e, :inner))
        return _node!(T, get(() -> error("to_blob: unary operator $(_fn_of(e)) has no opcode"), _UNOP, _fn_of(e)), a)
    elseif n == :Node2
        a = _walk!(T, getfield(e, :inner1)); b = _walk!(T, getfield(e, :inner2))
        return _node!(T, get(() -> error("to_blob: binary operator $(_fn_of(e)) has no opcode"), _BINOP, _fn_of(e)), a, b)
    end
    error("to_blob: unsupported expression node $(typeof(e))")
end

# linked list (newest first, `.inner` = previous) → call order
function _chain(head)
    out = Any[]
    while hasfield(typeof(head), :f) && hasfield(typeof(head), :itr)
        pushfirst!(out, head); head = getfield(head, :inner)
    end
    return out
end

# `slabs`  (optional, needed by iem_create_sharded): one `(offset0, dims, groups)` per `add_var` slab in
#          creation order — what `data.infvar_mappings` / `finvar_mappings` hold (`Variable.offset/.size`,
#          src/infiniteopt_backend.jl:476-479) plus the parameter-group index of each axis (0 = none);
# `template_groups` (optional): the parameter-group index of each template's (1-D) iterator in
#          add_obj/add_con call order, 0 = none — becomes the grid hint the window cut keys on.
# UNEXECUTED like the rest of this file.  Product iterators (pandemic's t x xi) reach the library as
# flat lists and are recovered as lattices without group ids: sharding them needs the Python producer.
function to_blob(core; slabs = nothing, template_groups = nothing)::Vector{UInt8}
    A = _Arrays(NTuple{5,Any}[], Dict{Any,Int}())
    core_ids = [_add_array!(A, Array(core.x0)), _add_array!(A, Array(core.lvar)),
                _add_array!(A, Array(core.uvar)), _add_array!(A, Array(core.θ))]
    lcon, ucon = Array(core.lcon), Array(core.ucon)
    objs, cons = _chain(core.obj), _chain(core.con)
    # add_obj/add_con CALL order decides the Hessian block offsets: merge the two lists by o2
    order = Tuple{Int,Any}[]
    i = j = 1
    while i <= length(objs) || j <= length(cons)
        takeobj = j > length(cons) || (i <= length(objs) && objs[i].f.o2 <= cons[j].f.o2)
        takeobj ? (push!(order, (0, objs[i])); i += 1) : (push!(order, (1, cons[j])); j += 1)
    end
    tpls = Vector{Vector{UInt64}}()
    for (kind, t) in order
        items = collect(Any, t.itr)
        T = _Tpl(items, A, Dict{Any,Int}(), Dict{Any,Int}(), [], [], [], NTuple{4,UInt64}[], nothing)
        root = _walk!(T, t.f.f)
        n = length(items)
        gid, origin = T.first_ap === nothing ? (n == 1 ? (0, 0) : (-1, 0)) : (4096 + 1, T.first_ap)
        if template_groups !== nothing && T.first_ap !== nothing && template_groups[length(tpls) + 1] > 0
            gid = template_groups[length(tpls) + 1] + 1          # one digit, base 4096: (group + 1)
        end
        w = UInt64[_w(kind), _w(n), _w(1), _w(n), _w(1), _w(1), _w(gid), _w(origin), _w(0), _w(0),
                   _w(length(T.ifields)), _w(length(T.ffields)), _w(length(T.idx)), _w(length(T.nodes)), _w(root)]
        if kind == 1
            rows = (t.f.o0 + 1):(t.f.o0 + n)
            append!(w, [_w(1), _w(0), _w(_add_array!(A, lcon[rows])), _w(1), _w(0), _w(_add_array!(A, ucon[rows]))])
        else
            append!(w, [_w(0), _w(0.0), _w(-1), _w(0), _w(0.0), _w(-1)])
        end
        foreach(f -> append!(w, f), T.ifields); foreach(f -> append!(w, f), T.ffields)
        foreach(x -> append!(w, x), T.idx)
        for nd in T.nodes; append!(w, nd); end
        push!(tpls, w)
    end
    narr, ntpl = length(A.recs), length(tpls)
    pos = 14 + 6 * narr + ntpl
    tpl_off = Int[]; for w in tpls; push!(tpl_off, pos); pos += length(w); end
    slab_words = UInt64[]
    if slabs !== nothing                                       # include/iem_blob.h: n, then n x {off, nd, dims[3], group[3]}
        push!(slab_words, _w(length(slabs)))
        for (off, dims, groups) in slabs
            nd = length(dims)
            append!(slab_words, _w.([off, nd, dims..., ones(Int, 3 - nd)..., groups..., zeros(Int, 3 - nd)...]))
        end
    end
    slab_off = isempty(slab_words) ? 0 : pos
    pos += length(slab_words)
    arr_off = Int[]; for r in A.recs; push!(arr_off, r[3] === nothing ? 0 : pos); r[3] === nothing || (pos += r[2]); end
    head = UInt64[_w(0x31424f4c424d4549), _w(1), _w(core.nvar), _w(core.npar), _w(core.ncon), _w(ntpl), _w(narr),
                  _w(core.minimize ? 1 : 0), _w(pos), _w(slab_off), _w.(core_ids)...]
    for (r, off) in zip(A.recs, arr_off); append!(head, [_w(r[1]), _w(r[2]), _w(off), r[4], r[5], _w(0)]); end
    append!(head, _w.(tpl_off)); foreach(w -> append!(head, w), tpls)
    append!(head, slab_words)
    for r in A.recs
        r[3] === nothing || append!(head, reinterpret(UInt64, r[3]))
    end
    @assert length(head) == pos
    return collect(reinterpret(UInt8, head))
end

end # module
