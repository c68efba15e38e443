from egopose_amd.agent import AgentEgo  # noqa: F401
