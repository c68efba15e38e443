from egopose_amd.rl_core import LoggerRL  # noqa: F401
