"""`from rl_coach.graph_managers.graph_manager import ScheduleParameters` — the import path presets use
(graph_manager.py:40-69); the class lives beside the manager that consumes it."""
from .basic_rl_graph_manager import ScheduleParameters  # noqa: F401
