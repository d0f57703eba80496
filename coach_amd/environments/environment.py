"""Level selection and the environment parameter base — rl_coach/environments/environment.py:36-110
(`LevelSelection`, `SingleLevelSelection`, `EnvironmentParameters`): what a preset's `env_params` is made of."""
from ..base_parameters import Parameters


class LevelSelection(object):
    def __init__(self, level):
        self.selected_level = level

    def select(self, level):
        self.selected_level = level

    def __str__(self):
        if self.selected_level is None:
            raise ValueError("No level has been selected. Please select a level using the -lvl command line flag, "
                             "or change the level in the preset.")
        return self.selected_level


class SingleLevelSelection(LevelSelection):
    def __init__(self, levels, force_lower=True):
        super().__init__(None)
        self.levels = levels
        if isinstance(levels, list):
            self.levels = {level: level for level in levels}
        if isinstance(levels, str):
            self.levels = {levels: levels}
        self.force_lower = force_lower

    def __str__(self):
        if self.selected_level is None:
            raise ValueError("No level has been selected. Please select a level using the -lvl command line flag, "
                             "or change the level in the preset. \nThe available levels are: \n{}"
                             .format(', '.join(sorted(self.levels.keys()))))
        selected_level = self.selected_level.lower() if self.force_lower else self.selected_level
        if selected_level not in self.levels.keys():
            raise ValueError("The selected level ({}) is not part of the available levels ({})"
                             .format(selected_level, ', '.join(self.levels.keys())))
        return self.levels[selected_level]


class EnvironmentParameters(Parameters):
    def __init__(self, level=None):
        self.level = level
        self.frame_skip = 4
        self.seed = None
        self.human_control = False
        self.custom_reward_threshold = None
        self.default_input_filter = None
        self.default_output_filter = None
        self.experiment_path = None
        self.target_success_rate = 1.0

    @property
    def path(self):
        return 'coach_amd.environments.environment:Environment'
