"""`rl_coach.graph_managers.graph_manager` as far as presets touch it: the schedule parameter classes
(ScheduleParameters graph_manager.py:43-49, SimpleScheduleWithoutEvaluation :60-66, SimpleSchedule :69-78).  The manager
that consumes a schedule is basic_rl_graph_manager.BasicRLGraphManager (heatup, then train_and_act / evaluate periods
until improve_steps); its ``improve()`` reads exactly these four fields."""
from ..core_types import EnvironmentEpisodes, EnvironmentSteps, TrainingSteps

_FOREVER = 10000000000


class ScheduleParameters(object):
    """The reference's base class leaves the four fields None (:43-49) and every preset fills them; here the base
    already carries the values the reference's own "no evaluation" schedule has, so a preset that sets only some fields
    still runs."""

    def __init__(self):
        self.heatup_steps = EnvironmentSteps(0)
        self.evaluation_steps = EnvironmentEpisodes(0)
        self.steps_between_evaluation_periods = EnvironmentSteps(10000)
        self.improve_steps = TrainingSteps(_FOREVER)


class SimpleScheduleWithoutEvaluation(ScheduleParameters):      # :60-66
    def __init__(self, improve_steps=None):
        super().__init__()
        self.improve_steps = improve_steps if improve_steps is not None else TrainingSteps(_FOREVER)
        self.steps_between_evaluation_periods = self.improve_steps   # one period: the evaluation never comes due


class SimpleSchedule(ScheduleParameters):                       # :69-78
    def __init__(self, improve_steps=None, steps_between_evaluation_periods=None, evaluation_steps=None):
        super().__init__()
        self.improve_steps = improve_steps if improve_steps is not None else TrainingSteps(_FOREVER)
        self.steps_between_evaluation_periods = steps_between_evaluation_periods \
            if steps_between_evaluation_periods is not None else EnvironmentEpisodes(50)
        self.evaluation_steps = evaluation_steps if evaluation_steps is not None else EnvironmentEpisodes(5)
