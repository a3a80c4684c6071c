# dump_reference.jl — UNEXECUTED here (no `julia` in the build image).  Run in a real session
# of the reference to PIN parity: it writes, for one transcribed model, the blob this build
# consumes plus everything ExaModels itself computes at a seeded point, as raw little-endian
# files that tests/test_reference_dump.py compares against (oracle on CPU, HIP path on GPU).
#
#     using InfiniteOpt, InfiniteExaModels, NLPModels, Random
#     include("MI355XBackend.jl"); include("dump_reference.jl")
#     im = ... build an InfiniteModel with ExaTranscriptionBackend (backend = nothing) ...
#     build_transformation_backend!(im)                       # src/infiniteopt_backend.jl:150-160
#     tb = transformation_backend(im)
#     dump_reference(tb.core, tb.model, "tests/golden/reference/quadrotor_100")
#
# Files: model.blob, x.f64, y.f64, obj.f64, grad.f64, cons.f64, jac_{rows,cols}.i64,
# jac_vals.f64, hess_{rows,cols}.i64, hess_vals.f64, meta.txt ("key value" lines).
using NLPModels, Random

function dump_reference(core, model, dir::AbstractString; seed::Int = 0, obj_weight::Float64 = 1.0)
    mkpath(dir)
    put(name, v) = open(io -> write(io, v), joinpath(dir, name), "w")
    put("model.blob", MI355X.to_blob(core))
    m = model.meta
    rng = MersenneTwister(seed)
    x = Array(m.x0) .+ 0.1 .* randn(rng, m.nvar)
    x .= clamp.(x, max.(Array(m.lvar), -1e3), min.(Array(m.uvar), 1e3))
    y = randn(rng, m.ncon)
    put("x.f64", x); put("y.f64", y)
    put("obj.f64", [NLPModels.obj(model, x)])
    put("grad.f64", NLPModels.grad(model, x))
    put("cons.f64", NLPModels.cons(model, x))
    jr, jc = NLPModels.jac_structure(model)
    put("jac_rows.i64", collect(Int64, jr)); put("jac_cols.i64", collect(Int64, jc))
    put("jac_vals.f64", NLPModels.jac_coord(model, x))
    hr, hc = NLPModels.hess_structure(model)
    put("hess_rows.i64", collect(Int64, hr)); put("hess_cols.i64", collect(Int64, hc))
    put("hess_vals.f64", NLPModels.hess_coord(model, x, y; obj_weight = obj_weight))
    open(joinpath(dir, "meta.txt"), "w") do io
        println(io, "nvar ", m.nvar); println(io, "ncon ", m.ncon)
        println(io, "nnzj ", m.nnzj); println(io, "nnzh ", m.nnzh)
        println(io, "obj_weight ", obj_weight); println(io, "index_base 1")
        println(io, "examodels ", string(pkgversion(parentmodule(typeof(model)))))
    end
    return dir
end
