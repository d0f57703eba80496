"""Schedule driver for one vector agent + one vector environment — the scheduling semantics of
rl_coach/graph_managers/graph_manager.py (heatup :400-424, train_and_act :463-489, evaluate
:491-523, improve :525-556) and basic_rl_graph_manager.py:35-78 (one agent, one environment, built
from `Parameters.path = 'module:Class'`, utils.py:389-404), with the CSV signal columns of
logger.py:189-335 / agent.py:509-556 written once per completed episode group.

Only what the hot path needs is here: no TF sessions, checkpoints, Redis/S3, dashboard or CLI.
"""
import csv
import importlib
import time

from ..core_types import EnvironmentEpisodes, EnvironmentSteps, RunPhase, TrainingSteps


class ScheduleParameters(object):                        # base_parameters.py:589-601
    def __init__(self):
        self.heatup_steps = EnvironmentSteps(0)
        self.evaluation_steps = EnvironmentEpisodes(0)
        self.steps_between_evaluation_periods = EnvironmentSteps(10000)
        self.improve_steps = TrainingSteps(10000000000)


def dynamic_import(path):
    """'module.path:ClassName' -> class (utils.py:389-404)."""
    module, cls = path.split(":")
    return getattr(importlib.import_module(module), cls)


class CsvLogger(object):
    """The columns of the reference's experiment CSV that the hot path produces (logger.py:291-335)."""
    COLUMNS = ["Episode #", "Training Iter", "In Heatup", "ER #Transitions", "Total steps", "Epsilon",
               "Training Reward", "Evaluation Reward", "Episode Length", "Update Target Network",
               "Loss/Mean", "Wall-Clock Time"]

    def __init__(self, path=None):
        self.path, self.rows = path, []
        self._t0 = time.time()
        if path:
            with open(path, "w", newline="") as f:
                csv.writer(f).writerow(self.COLUMNS)

    def write(self, **kw):
        kw["Wall-Clock Time"] = time.time() - self._t0
        row = [kw.get(c, "") for c in self.COLUMNS]
        self.rows.append(dict(zip(self.COLUMNS, row)))
        if self.path:
            with open(self.path, "a", newline="") as f:
                csv.writer(f).writerow(row)


class BasicRLGraphManager(object):
    def __init__(self, agent_params, env_params, schedule_params, vis_params=None, preset_validation_params=None,
                 name='simple_rl_graph', device=None, dist=None, csv_path=None):
        """Reference signature (basic_rl_graph_manager.py:35-42: agent_params, env_params, schedule_params,
        vis_params, preset_validation_params, name) + where to run (device, dist) and where to log."""
        import torch
        from ..base_parameters import PresetValidationParameters, VisualizationParameters
        self.agent_params, self.env_params, self.schedule = agent_params, env_params, schedule_params
        self.schedule_params = schedule_params
        self.visualization_parameters = vis_params or VisualizationParameters()
        self.preset_validation_params = preset_validation_params or PresetValidationParameters()
        self.name = name
        self.device = device or torch.device("cuda", 0)
        self.dist = dist
        self.environment = self.agent = None
        self.logger = CsvLogger(csv_path)
        self.total_steps_counters = {RunPhase.HEATUP: 0, RunPhase.TRAIN: 0, RunPhase.TEST: 0}
        self.training_steps = 0
        self.phase = RunPhase.UNDEFINED
        self._episodes_logged = 0

    # ------------------------------------------------------------------ graph creation (:62-78)
    def create_graph(self):
        rank = self.dist.rank if self.dist is not None else 0
        from ..compat import resolve_reference_style
        resolve_reference_style(self.agent_params, self.env_params)
        self.environment = dynamic_import(self.env_params.path)(self.env_params, self.device, rank=rank)
        self.agent = dynamic_import(self.agent_params.path)(self.agent_params, self.environment, self.device,
                                                            dist=self.dist)
        return self

    def verify_graph_was_created(self):
        if self.agent is None:
            self.create_graph()

    def _set_phase(self, phase):
        self.phase = phase
        self.agent.phase = phase

    # --------------------------------------------------------------------------------- stepping
    def _act(self):
        """One vector step; returns the env-steps it advanced (GraphManager.act, :426-461)."""
        self.agent.act()
        n = self.agent.n_env
        self.total_steps_counters[self.phase] += n
        self._log_finished_episodes()
        return n

    def _log_finished_episodes(self):
        st = None
        done_groups = 0
        L = getattr(self.agent, "L", None)
        steps = self.total_steps_counters[RunPhase.HEATUP] + self.total_steps_counters[RunPhase.TRAIN]
        if L:
            done_groups = steps // (L * self.agent.n_env)
        if done_groups > self._episodes_logged and self.phase != RunPhase.TEST:
            st = self.agent.episode_statistics()
            self._episodes_logged = done_groups
            sig = self.agent.signals
            loss = next((float(sig[k]) for k in ("Loss", "Surrogate loss") if k in sig), "")
            eps = ""
            if hasattr(self.agent, "exploration_policy") and hasattr(self.agent.exploration_policy, "epsilon"):
                eps = self.agent.exploration_policy.epsilon()
            self.logger.write(**{"Episode #": st["episodes"], "Training Iter": self.agent.training_iteration,
                                 "In Heatup": int(self.phase == RunPhase.HEATUP),
                                 "ER #Transitions": self.agent.memory.num_transitions(),
                                 "Total steps": self.agent.total_steps_counter, "Epsilon": eps,
                                 "Training Reward": st["mean_return"], "Episode Length": st["mean_length"],
                                 "Loss/Mean": loss})

    def heatup(self, steps):                                                   # :400-424
        self.verify_graph_was_created()
        if steps.num_steps > 0:
            self._set_phase(RunPhase.HEATUP)
            done = 0
            while done < steps.num_steps:
                done += self._act()

    def train_and_act(self, steps):                                            # :463-489
        self.verify_graph_was_created()
        self._set_phase(RunPhase.TRAIN)
        done = 0
        while done < steps.num_steps:
            done += self._act()
            before = self.agent.training_iteration
            self.agent.train()
            self.training_steps += self.agent.training_iteration - before

    def evaluate(self, steps):                                                 # :491-523
        """Act greedily for `steps` episodes per env without storing or training; returns the mean
        evaluation reward (the success-rate early exit of the reference is not used here)."""
        self.verify_graph_was_created()
        if steps.num_steps <= 0 or not hasattr(self.agent, "evaluate_episodes"):
            return None
        self._set_phase(RunPhase.TEST)
        before = self.environment.total_steps
        reward = self.agent.evaluate_episodes(steps.num_steps)
        self.total_steps_counters[RunPhase.TEST] += self.environment.total_steps - before
        self.logger.write(**{"Episode #": self._episodes_logged, "Training Iter": self.agent.training_iteration,
                             "In Heatup": 0, "Total steps": self.agent.total_steps_counter,
                             "Evaluation Reward": reward})
        return reward

    def improve(self):                                                         # :525-556
        self.verify_graph_was_created()
        self.heatup(self.schedule.heatup_steps)
        imp = self.schedule.improve_steps
        counter = (lambda: self.training_steps) if isinstance(imp, TrainingSteps) else \
            (lambda: self.total_steps_counters[RunPhase.TRAIN])
        count_end = counter() + imp.num_steps
        while counter() < count_end:
            self.train_and_act(self.schedule.steps_between_evaluation_periods)
            self.evaluate(self.schedule.evaluation_steps)
        self.agent.check_status() if hasattr(self.agent, "check_status") else None
        return self.logger.rows
