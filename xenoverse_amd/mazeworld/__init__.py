"""MazeWorld on MI355X.  Drop-in for the step/reset/observation path of `xenoverse.mazeworld`."""
from .tables import DEFAULT_ACTION_SPACE_16, DEFAULT_ACTION_SPACE_32, build_tables  # noqa: F401
from .textures import REFERENCE_TEXTURE_COUNTS, load_texture_library, make_texture_library, texture_counts  # noqa: F401
from .vec_env import MazeWorldVecEnv  # noqa: F401
from .task_sampler import MazeTaskSampler, Resampler  # noqa: F401
from .agents import AgentBase, OracleAgent, SmartSLAMAgent, teacher_rollout  # noqa: F401
