"""Domain-randomised classic control on MI355X.  Drop-in for `xenoverse.metacontrol` random-cartpole / random-acrobot."""
from .acrobot import AcrobotVecEnv, sample_acrobot  # noqa: F401
from .cartpole import CartPoleVecEnv, sample_cartpole  # noqa: F401
