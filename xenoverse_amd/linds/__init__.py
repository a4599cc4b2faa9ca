"""LinDS on MI355X.  Drop-in for the step/reset path of `xenoverse.linds` (reference package)."""
from .tables import build_dynamics_matrices, build_tables  # noqa: F401
from .vec_env import LinDSVecEnv, pad_tables  # noqa: F401
from .task_sampler import LinearDSSampler, LinearDSSamplerRandomDim, RandomFourier  # noqa: F401
