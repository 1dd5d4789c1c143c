# BALNLPModelsHIP.jl -- drop-in for src/BALNLPModels.jl (+ src/ReadFiles.jl, src/JacobianByHand.jl) of
# CelestineAngla/BundleAdjustment.jl: same type name, fields, constructor and NLPModels methods, the bodies are ccalls
# into libba_hip.so (MI355X).  Usage: in src/solve_ba.jl:1 replace `include("BALNLPModels.jl")` by
# `include("<this repo>/julia/BALNLPModelsHIP.jl")`; `FeasibilityResidual(BA)` (src/solve_ba.jl:25) then works unchanged,
# it only calls cons!, jac_structure! and jac_coord!.
using NLPModels
include("BALHIP.jl")

"readfile(filename, T): src/ReadFiles.jl:9-53 (same `<repo>/Data/<filename>` convention; BA_DATA_DIR overrides)"
function readfile(filename :: String, T :: Type = Float64)
  filepath = joinpath(get(ENV, "BA_DATA_DIR", joinpath(@__DIR__, "..", "Data")), filename)
  nc, np, no = Ref{Int64}(0), Ref{Int64}(0), Ref{Int64}(0)
  bacheck(ccall((:ba_read_bal_header, libba), Cint, (Cstring, Ref{Int64}, Ref{Int64}, Ref{Int64}), filepath, nc, np, no))
  ncams, npnts, nobs = Int(nc[]), Int(np[]), Int(no[])
  @info "$filename: reading" ncams npnts nobs
  cam_indices = Vector{Int}(undef, nobs)
  pnt_indices = Vector{Int}(undef, nobs)
  pt2d = Vector{T}(undef, 2 * nobs)
  x0 = Vector{T}(undef, 3 * npnts + 9 * ncams)
  if T == Float32
    bacheck(ccall((:ba_read_bal_f32, libba), Cint,
                  (Cstring, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float32}, Ptr{Float32}),
                  filepath, ncams, npnts, nobs, cam_indices, pnt_indices, pt2d, x0))
  elseif T == Float64
    bacheck(ccall((:ba_read_bal, libba), Cint,
                  (Cstring, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}),
                  filepath, ncams, npnts, nobs, cam_indices, pnt_indices, pt2d, x0))
  else
    error("readfile: T must be Float64 or Float32 on the HIP path")
  end
  return cam_indices, pnt_indices, pt2d, x0, ncams, npnts, nobs
end

"name(filename): src/BALNLPModels.jl:58-68, verbatim behaviour (\"LadyBug/problem-49-7776-pre.txt.bz2\" -> \"LadyBug-49-7776\")"
function name(filename :: AbstractString)
  k = findfirst(isequal('/'), filename)
  l = k + 8
  while filename[l] != 'p'
    l += 1
  end
  return filename[1 : k - 1] * filename[k + 8 : l - 2]
end

# src/BALNLPModels.jl:79-88 plus the device handle
mutable struct BALNLPModel <: AbstractNLPModel
  meta :: NLPModelMeta
  counters :: Counters
  cams_indices :: Vector{Int}
  pnts_indices :: Vector{Int}
  pt2d :: AbstractVector
  nobs :: Int
  npnts :: Int
  ncams :: Int
  handle :: Ptr{Cvoid}
end

function BALNLPModel(filename :: AbstractString, T :: Type = Float64; device :: Integer = 0)   # src/BALNLPModels.jl:91-106
  cams_indices, pnts_indices, pt2d, x0, ncams, npnts, nobs = readfile(String(filename), T)
  nvar = 9 * ncams + 3 * npnts
  ncon = 2 * nobs
  meta = NLPModelMeta(nvar, ncon=ncon, x0=x0, lcon=fill(0.0, ncon), ucon=fill(0.0, ncon), nnzj=2 * nobs * 12,
                      name=name(filename))
  @info "BALNLPModel $filename" nvar ncon
  h = Ref{Ptr{Cvoid}}(C_NULL)
  pt2d64 = Float64.(pt2d)   # the device keeps Float64 and Float32 mirrors of the observations
  bacheck(ccall((:ba_problem_create, libba), Cint,
                (Cint, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ref{Ptr{Cvoid}}),
                device, ncams, npnts, nobs, cams_indices, pnts_indices, pt2d64, h))
  nlp = BALNLPModel(meta, Counters(), cams_indices, pnts_indices, pt2d, nobs, npnts, ncams, h[])
  finalizer(m -> (m.handle != C_NULL && ccall((:ba_problem_destroy, libba), Cvoid, (Ptr{Cvoid},), m.handle);
                  m.handle = C_NULL), nlp)
  return nlp
end

NLPModels.obj(model :: BALNLPModel, x :: AbstractVector) = 0.0                              # :109
NLPModels.grad!(model :: BALNLPModel, x :: AbstractVector, g :: AbstractVector) = fill!(g, 0)   # :112

# cons!(nlp, x, cx): src/BALNLPModels.jl:115-122 (residuals! :39-55, projection! :17-33)
function NLPModels.cons!(nlp :: BALNLPModel, x :: Vector{Float64}, cx :: Vector{Float64})
  increment!(nlp, :neval_cons)
  bacheck(ccall((:ba_residual, libba), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), nlp.handle, x, cx))
  return cx
end
function NLPModels.cons!(nlp :: BALNLPModel, x :: Vector{Float32}, cx :: Vector{Float32})
  increment!(nlp, :neval_cons)
  bacheck(ccall((:ba_residual_f32, libba), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}), nlp.handle, x, cx))
  return cx
end
# any other AbstractVector (views, mixed element types): through dense copies, like the reference's generic code would
function NLPModels.cons!(nlp :: BALNLPModel, x :: AbstractVector, cx :: AbstractVector)
  T = eltype(x) == Float32 ? Float32 : Float64
  tmp = Vector{T}(undef, length(cx))
  NLPModels.cons!(nlp, Vector{T}(x), tmp)
  cx .= tmp
  return cx
end

# jac_structure!(nlp, rows, cols): src/BALNLPModels.jl:125-158
function NLPModels.jac_structure!(nlp :: BALNLPModel, rows :: Vector{Int}, cols :: Vector{Int})
  increment!(nlp, :neval_jac)
  bacheck(ccall((:ba_jac_structure, libba), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}), nlp.handle, rows, cols))
  return rows, cols
end
function NLPModels.jac_structure!(nlp :: BALNLPModel, rows :: AbstractVector{<:Integer}, cols :: AbstractVector{<:Integer})
  r, c = Vector{Int}(undef, length(rows)), Vector{Int}(undef, length(cols))
  NLPModels.jac_structure!(nlp, r, c)
  rows .= r
  cols .= c
  return rows, cols
end

# jac_coord!(nlp, x, vals): src/BALNLPModels.jl:161-206 + src/JacobianByHand.jl:5-101 (NaN -> 0, :201)
function NLPModels.jac_coord!(nlp :: BALNLPModel, x :: Vector{Float64}, vals :: Vector{Float64})
  increment!(nlp, :neval_jac)
  bacheck(ccall((:ba_jac_coord, libba), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), nlp.handle, x, vals))
  return vals
end
function NLPModels.jac_coord!(nlp :: BALNLPModel, x :: Vector{Float32}, vals :: Vector{Float32})
  increment!(nlp, :neval_jac)
  bacheck(ccall((:ba_jac_coord_f32, libba), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}), nlp.handle, x, vals))
  return vals
end
function NLPModels.jac_coord!(nlp :: BALNLPModel, x :: AbstractVector, vals :: AbstractVector)
  T = eltype(x) == Float32 ? Float32 : Float64
  tmp = Vector{T}(undef, length(vals))
  NLPModels.jac_coord!(nlp, Vector{T}(x), tmp)
  vals .= tmp
  return vals
end

"Jᵀr from the COO Jacobian (mul_sparse with swapped indices, src/lma_aux.jl:194-212 as used at src/lm.jl:57,370)"
function jtr!(nlp :: BALNLPModel, vals :: Vector{Float64}, r :: Vector{Float64}, out :: Vector{Float64})
  bacheck(ccall((:ba_jtr, libba), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), nlp.handle, vals, r, out))
  return out
end
