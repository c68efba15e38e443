from egopose_amd.rl_core import TrajBatchEgo  # noqa: F401
