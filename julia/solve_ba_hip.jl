# src/solve_ba.jl:1-27 with the two `include` lines pointed at the HIP shim; everything else is the reference's script.
#   julia solve_ba_hip.jl LadyBug/problem-49-7776-pre.txt.bz2 LDL AMD None
include("BALNLPModelsHIP.jl")
include("LevenbergMarquardtHIP.jl")

facto = ARGS[2] == "QR" ? :QR : :LDL
perm = ARGS[3] == "Metis" ? :Metis : :AMD
norm = ARGS[4] == "A" ? :A : ARGS[4] == "J" ? :J : :None

BA = BALNLPModel(ARGS[1])
fr_BA = FeasibilityResidual(BA)
stats = Levenberg_Marquardt(fr_BA, facto, perm, norm)
print("\n ------------ \nStats : \n", stats)
