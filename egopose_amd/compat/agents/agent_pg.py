from egopose_amd.agent import AgentPG  # noqa: F401
