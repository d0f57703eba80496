"""`rl_coach.agents.ddqn_agent` import path (ddqn_agent.py:24-43): Double DQN lives with DQN here."""
from .dqn_agent import DDQNAgent, DDQNAgentParameters  # noqa: F401
