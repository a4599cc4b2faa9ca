"""AnyMDP on MI355X.  Drop-in for the step/reset path of `xenoverse.anymdp` (reference package)."""
from .tables import build_obs_tables, build_tables, from_blocked, row_cdf, row_lines, to_blocked, validate_task  # noqa: F401
from .vec_env import AnyMDPVecEnv  # noqa: F401
from .task_sampler import (AnyMDPTaskSampler, AnyPOMDPTaskSampler, GarnetTaskSampler,  # noqa: F401
                           MultiTokensAnyPOMDPTaskSampler)
from .device_sampler import sample_tasks_device  # noqa: F401,E402
