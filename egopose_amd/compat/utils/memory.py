from egopose_amd.rl_core import Memory  # noqa: F401
