# BALHIP.jl -- shared part of the Julia shim over libba_hip.so (include/ba_hip.h): library handle, error mapping and the
# C structs.  Included by BALNLPModelsHIP.jl and LevenbergMarquardtHIP.jl, which replace src/BALNLPModels.jl and
# src/LevenbergMarquardt.jl / src/lm.jl of CelestineAngla/BundleAdjustment.jl in the `include` lines of
# src/solve_ba.jl:1-2, src/main.jl, src/benchmark.jl, src/diffprecsions.jl and src/benchmark_diffprec.jl.
#
# NOT EXECUTED in the build image (no Julia runtime there): the struct layouts below are kept equal to the C header by
# tests/test_host.py::test_julia_shim_matches_header, which reads THIS file's field lists and compares them with the
# sizeof/offsetof the C compiler reports (tests/c_abi/abi_check.c).

const libba = get(ENV, "BA_HIP_LIB", joinpath(@__DIR__, "..", "bundleadjustment.jl_amd", "libba_hip.so"))

# error codes of include/ba_hip.h
const BA_OK = 0
const BA_ERR_ZERO_PIVOT = 4

"Zero pivot in the LDLᵀ factorisation: src/ldl_aux.jl:45-47,199"
struct SQDException <: Exception
  msg :: String
end

struct BAError <: Exception
  code :: Int
  msg :: String
end

function bacheck(rc :: Integer)
  rc == BA_OK && return nothing
  msg = unsafe_string(ccall((:ba_last_error, libba), Cstring, ()))
  rc == BA_ERR_ZERO_PIVOT && throw(SQDException(msg))
  throw(BAError(rc, msg))
end

# struct ba_lm_opts (include/ba_hip.h) -- field order and types ARE the ABI
struct BaLmOpts
  variant :: Cint
  facto :: Cint
  normalize :: Cint
  linesearch :: Cint
  facto_type :: Cint
  ite_max :: Cint
  verbose :: Cint
  x_f32 :: Cint
  restol :: Cdouble
  satol :: Cdouble
  srtol :: Cdouble
  oatol :: Cdouble
  ortol :: Cdouble
  atol :: Cdouble
  rtol :: Cdouble
  nu_d :: Cdouble
  nu_m :: Cdouble
  lambda :: Cdouble
  delta_d :: Cdouble
  max_time :: Cdouble
  pcg_tol :: Cdouble
  pcg_max_iter :: Cint
  perm :: Cint
end

# struct ba_lm_stats (include/ba_hip.h)
mutable struct BaLmStats
  status :: Cint
  iter :: Cint
  n_accepted :: Cint
  n_rejected :: Cint
  n_residual :: Cint
  n_jacobian :: Cint
  n_factor :: Cint
  n_cg :: Cint
  objective :: Cdouble
  dual_feas :: Cdouble
  lambda_final :: Cdouble
  elapsed_s :: Cdouble
  loop_s :: Cdouble
  BaLmStats() = new(-1, 0, 0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0)
end

# BA_ST_* -> the status symbols of src/lm.jl:391-405 (index = code + 1)
const BA_STATUS = (:small_step, :first_order, :small_residual, :acceptable, :neg_pred, :exception, :max_iter)
