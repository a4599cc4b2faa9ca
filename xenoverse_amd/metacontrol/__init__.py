"""Domain-randomised classic control on MI355X.  Drop-in for `xenoverse.metacontrol` random-cartpole."""
from .cartpole import CartPoleVecEnv, sample_cartpole  # noqa: F401
