from egopose_amd.agent import Agent  # noqa: F401
