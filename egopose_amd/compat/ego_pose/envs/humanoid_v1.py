from egopose_amd.env import HumanoidEnv  # noqa: F401
