"""bundleadjustment.jl_amd -- MI355X (gfx950) implementation of the hot path of CelestineAngla/BundleAdjustment.jl:
reprojection residual, hand-derived Jacobian and the Levenberg-Marquardt step, behind the reference's
NLPModels-style surface.  The compute lives in libba_hip.so (csrc/, C ABI in include/ba_hip.h); this package is the
host-side mirror of the reference interface.  The directory name contains a dot, so it is loaded through
`__graft_entry__.load_package()` (importlib) under the module name `bundleadjustment_jl_amd`.
"""
# torch ships its own libamdhip64.so; whichever HIP runtime is loaded first serves the whole process (same SONAME).
# Load torch's first when torch is installed, so that torch.cuda / torch.distributed keep working next to libba_hip.so
# (loaded the other way round torch reports "No HIP GPUs are available").  The C ABI itself does not need torch.
try:
    import torch as _torch  # noqa: F401
except ImportError:  # pragma: no cover
    _torch = None
from . import _lib
from ._lib import BAError, SQDException, device_count
from .lm import GenericExecutionStats, Levenberg_Marquardt, lm_step, schur_pattern, schur_memory, set_ordering, schur_ordering_used
from ._lib import schur_ordering
from .model import BALNLPModel, FeasibilityResidual
from .readfiles import name, readfile
from . import synthetic
from . import parallel

__all__ = ["BALNLPModel", "FeasibilityResidual", "Levenberg_Marquardt", "GenericExecutionStats", "readfile", "name",
           "BAError", "SQDException", "device_count", "synthetic", "parallel", "lm_step", "schur_pattern", "schur_memory", "set_ordering", "schur_ordering_used",
           "schur_ordering"]
