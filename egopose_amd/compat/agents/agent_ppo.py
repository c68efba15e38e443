from egopose_amd.agent import AgentPPO  # noqa: F401
