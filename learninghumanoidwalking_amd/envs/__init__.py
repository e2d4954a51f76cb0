"""Host-side mirrors of the reference's environments (reference envs/__init__.py:13-19)."""
from .cartpole import CartpoleSpec, make_cartpole  # noqa: F401
from .h1 import H1Spec  # noqa: F401
from .h1_walk import H1WalkSpec  # noqa: F401
from .jvrc_step import JvrcStepSpec  # noqa: F401
from .jvrc_walk import JvrcWalkSpec  # noqa: F401

ENVIRONMENTS = {"cartpole": CartpoleSpec, "jvrc_walk": JvrcWalkSpec, "jvrc_step": JvrcStepSpec, "h1": H1Spec,
                "h1_walk": H1WalkSpec}


def single_env(name, **kw):
    """The reference's ``Env(path_to_yaml)`` single-env object for ``name`` (GPU required)."""
    from . import adapters
    return {"cartpole": adapters.CartpoleEnv, "jvrc_walk": adapters.JvrcWalkEnv, "jvrc_step": adapters.JvrcStepEnv,
            "h1": adapters.H1Env, "h1_walk": adapters.H1WalkEnv}[name](**kw)
