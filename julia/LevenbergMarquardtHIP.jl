# LevenbergMarquardtHIP.jl -- drop-in for src/LevenbergMarquardt.jl (4 positional arguments, what src/solve_ba.jl:26
# calls) AND src/lm.jl (5 positional arguments with `linesearch`: src/main.jl:30, src/diffprecsions.jl:41,
# src/benchmark_diffprec.jl:42-89).  Same function name, keywords and return type (GenericExecutionStats); the loop
# itself runs in libba_hip.so (ba_lm_solve), which follows the two reference loops line by line.
using NLPModels
using SolverTools
include("BALHIP.jl")

_tol(v) = v === nothing ? -1.0 : Float64(v)   # negative = "the variant's eps-derived default" (include/ba_hip.h)

# one log row per iteration, the reference's columns (src/lm.jl:120-121,304)
function _ba_log_row(ctx :: Ptr{Cvoid}, iter :: Cint, f :: Cdouble, df :: Cdouble, njtr :: Cdouble, lambda :: Cdouble,
                     ndelta :: Cdouble, rho :: Cdouble, acc :: Cint) :: Cvoid
  @info log_row(Any[Int(iter), f, df, njtr, lambda, ndelta, rho, acc != 0 ? "acc" : "rej"])
  return nothing
end

function _ba_lm(model, variant :: Int, facto :: Symbol, perm :: Symbol, normalize :: Symbol, linesearch :: Bool,
                x :: AbstractVector, facto_type :: DataType, restol, satol, srtol, oatol, ortol, atol, rtol, νd, νm, λ, δd,
                ite_max :: Int, max_time :: Real, pcg_tol :: Real = -1.0, pcg_max_iter :: Int = -1)
  # :PCG is an extension of the HIP path (no counterpart in the reference): matrix-free conjugate gradients on the reduced
  # camera system, include/ba_hip.h, ba_lm_opts.facto
  facto in (:QR, :LDL, :PCG) || error("facto must be :QR, :LDL or :PCG")
  (facto == :PCG && normalize != :None) && error("facto = :PCG has its own scaling (block-Jacobi preconditioner): normalize must be :None")
  (facto == :PCG && facto_type == Float32 && T != Float32) && error("facto = :PCG runs in Float64: facto_type = Float32 belongs to the direct branches")
  perm in (:AMD, :Metis) || error("perm must be :AMD or :Metis")   # src/lm.jl:84-88: orders the cameras of the reduced system (ba_lm_opts.perm)
  normalize in (:None, :J, :A) || error("normalize must be :None, :J or :A")
  nlp = model.nlp                       # the BALNLPModel inside FeasibilityResidual (src/solve_ba.jl:25)
  T = eltype(x)
  T in (Float64, Float32) || error("the HIP path iterates in Float64 or Float32")
  ft = facto_type == T ? 0 : facto_type == Float32 ? 1 : facto_type == Float16 ? 2 : error("facto_type must be Float64, Float32 or Float16")
  (T == Float32 && ft == 0 && variant == 1) && (ft = 1)   # eltype(x) = Float32: facto_type defaults to it (src/lm.jl:20)
  o = BaLmOpts(variant, facto == :QR ? 1 : facto == :PCG ? 2 : 0, normalize == :None ? 0 : normalize == :J ? 1 : 2, linesearch ? 1 : 0, ft,
               ite_max, 0, T == Float32 ? 1 : 0, _tol(restol), _tol(satol), _tol(srtol), _tol(oatol), _tol(ortol),
               _tol(atol), _tol(rtol), _tol(νd), _tol(νm), _tol(λ), _tol(δd), Float64(max_time), Float64(pcg_tol),
               Cint(pcg_max_iter), Cint(perm == :Metis ? 1 : 0))
  st = BaLmStats()
  xd = Vector{Float64}(x)               # the ABI carries the iterate as doubles (exact for Float32 values)
  cb = @cfunction(_ba_log_row, Cvoid, (Ptr{Cvoid}, Cint, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cint))
  @info log_header([:iter, :f, :Δf, :dFeas, :λ, :δ, :ρ, :status], [Int, T, T, T, T, T, T, String],
                   hdr_override = Dict(:f => "f(x)", :dFeas => "‖Jᵀr‖", :δ => "‖δ‖"))
  GC.@preserve xd st begin
    bacheck(ccall((:ba_lm_solve, libba), Cint, (Ptr{Cvoid}, Ref{BaLmOpts}, Ptr{Float64}, Ref{BaLmStats}, Ptr{Cvoid}, Ptr{Cvoid}),
                  nlp.handle, o, xd, st, cb, C_NULL))
  end
  x .= T.(xd)
  # the evaluations happened inside the library: keep the model's counters truthful (src/BALNLPModels.jl:116,126,162)
  nlp.counters.neval_cons += st.n_residual
  nlp.counters.neval_jac += st.n_jacobian + 1
  status = BA_STATUS[st.status + 1]
  if variant == 1   # src/lm.jl:409-415
    return GenericExecutionStats(status, model, solution=x, objective=st.objective, iter=Int(st.iter),
                                 elapsed_time=st.elapsed_s, dual_feas=st.dual_feas)
  else              # src/LevenbergMarquardt.jl:384: |Jᵀr| is reported as primal_feas there
    return GenericExecutionStats(status, model, solution=x, objective=st.objective, iter=Int(st.iter),
                                 elapsed_time=st.elapsed_s, primal_feas=st.dual_feas)
  end
end

"src/lm.jl:15-26 -- `Levenberg_Marquardt(model, facto, perm, normalize, linesearch; kwargs...)`"
function Levenberg_Marquardt(model :: AbstractNLSModel, facto :: Symbol, perm :: Symbol, normalize :: Symbol,
                             linesearch :: Bool;
                             x :: AbstractVector = copy(model.meta.x0), facto_type :: DataType = eltype(x),
                             restol = nothing, satol = nothing, srtol = nothing, oatol = nothing, ortol = nothing,
                             atol = nothing, rtol = nothing, νd = nothing, νm = nothing, λ = nothing, δd = nothing,
                             ite_max :: Int = 200, max_time :: Int = 3600, pcg_tol :: Real = -1.0, pcg_max_iter :: Int = -1)
  return _ba_lm(model, 1, facto, perm, normalize, linesearch, x, facto_type, restol, satol, srtol, oatol, ortol, atol, rtol,
                νd, νm, λ, δd, ite_max, max_time, pcg_tol, pcg_max_iter)
end

"src/LevenbergMarquardt.jl:16-26 -- the 4-argument method src/solve_ba.jl:26 calls (no linesearch, no facto_type)"
function Levenberg_Marquardt(model :: AbstractNLSModel, facto :: Symbol, perm :: Symbol, normalize :: Symbol;
                             x :: AbstractVector = copy(model.meta.x0),
                             restol = nothing, satol = nothing, srtol = nothing, oatol = nothing, ortol = nothing,
                             atol = nothing, rtol = nothing, νd = nothing, νm = nothing, λ = nothing, δd = nothing,
                             ite_max :: Int = 100)
  return _ba_lm(model, 0, facto, perm, normalize, false, x, eltype(x), restol, satol, srtol, oatol, ortol, atol, rtol,
                νd, νm, λ, δd, ite_max, 3600)
end
