"""Stand-in for legacy `gym` (the reference's maze_env imports `from gym import error, spaces, utils`)."""
from gymnasium import spaces, error, utils  # noqa: F401
from gymnasium import Env  # noqa: F401
