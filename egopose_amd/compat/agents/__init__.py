from agents.agent import Agent  # noqa: F401
from agents.agent_pg import AgentPG  # noqa: F401
from agents.agent_ppo import AgentPPO  # noqa: F401
