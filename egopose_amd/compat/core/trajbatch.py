from egopose_amd.rl_core import TrajBatch  # noqa: F401
