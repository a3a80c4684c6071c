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

# ---- blob writer ----------------------------------------------------------------------------
# Walks the host ExaCore (`core.obj` / `core.con` linked lists of SIMDFunction-carrying terms)
# and emits include/iem_blob.h.  Node mapping (ExaModels graph.jl types → opcodes):
#   Var{I}            → IEM_OP_VAR  + index expression (I an Int or a ParIndexed/Node2 ± Int tree)
#   ParameterNode{I}  → IEM_OP_PAR
#   ParIndexed{_, n}  → IEM_OP_DATA on item field n (float)  /  index field (int)
#   Real              → IEM_OP_CONST
#   Node1{F}          → unary opcode of F,  Node2{F} → binary opcode of F (Real operands → CONST)
# Item iterators (Vector{NamedTuple}) are written as explicit columns (mode GATHER); columns
# that are arithmetic progressions are written as AFFINE fields so that the generator can fuse
# templates over a support grid (grid_id = hash of the group aliases, origin = first index - 1).
function to_blob(core)::Vector{UInt8}
    error("to_blob: serialise `core` following include/iem_blob.h — see infiniteexamodels.jl_amd/core.py " *
          "(ExaCore.to_blob) for the reference implementation of the writer")
end

end # module
