"""Host-side mirrors of the reference's environments (reference envs/__init__.py:13-19)."""
from .cartpole import CartpoleSpec, make_cartpole  # noqa: F401
from .jvrc_walk import JvrcWalkSpec  # noqa: F401

ENVIRONMENTS = {"cartpole": CartpoleSpec, "jvrc_walk": JvrcWalkSpec}
