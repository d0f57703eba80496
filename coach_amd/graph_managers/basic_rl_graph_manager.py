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
    """The experiment CSV of the reference (logger.py:189-335): index column 'Episode #', the fixed columns
    Agent.update_log creates (agent.py:520-546, in that order) and four statistics columns per registered signal
    (:548-552).  tests/golden/csv_columns.json holds the column lists the real reference agents produce."""
    FIXED = ["Training Iter", "Epoch", "In Heatup", "ER #Transitions", "ER #Episodes", "Episode Length",
             "Total steps", "Epsilon", "Shaped Training Reward", "Training Reward", "Update Target Network",
             "Wall-Clock Time", "Evaluation Reward", "Shaped Evaluation Reward", "Success Rate",
             "Inverse Propensity Score", "Direct Method Reward", "Doubly Robust", "Weighted Importance Sampling",
             "Sequential Doubly Robust"]
    INDEX = "Episode #"

    def __init__(self, path=None, signal_names=()):
        self.path, self.rows = path, []
        self._t0 = None
        self.set_signals(signal_names)

    def set_signals(self, signal_names):
        self.COLUMNS = [self.INDEX] + self.FIXED + ["%s/%s" % (n, s) for n in signal_names
                                                    for s in ("Mean", "Stdev", "Max", "Min")]
        if self.path:
            with open(self.path, "w", newline="") as f:
                csv.writer(f).writerow(self.COLUMNS)

    def write(self, **kw):
        if self._t0 is None:
            self._t0 = time.time()                       # the wall clock starts with the first row (logger.py:240-246)
        kw.setdefault("Wall-Clock Time", time.time() - self._t0)
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
        if self.visualization_parameters.dump_csv and hasattr(self.agent, "enable_signal_statistics"):
            self.agent.enable_signal_statistics()
            self.logger.set_signals(self.agent.SIGNAL_NAMES)
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
        """Agent.update_log for every episode that finished on the last vector step (agent.py:509-556): one CSV row
        per episode — each env's episode is one 'Episode #' — with the episode's length and reward and the statistics
        of the signals accumulated since the previous row."""
        agent = self.agent
        if self.phase == RunPhase.TEST or not getattr(agent, "_episode_just_ended", False):
            return
        if getattr(agent, "signal_stats", None) is None:            # agents without the per-episode log (PPO)
            st = agent.episode_statistics()
            self._episodes_logged = st["episodes"]
            sig = agent.signals
            loss = next((float(sig[k]) for k in ("Loss", "Surrogate loss") if k in sig), "")
            self.logger.write(**{"Episode #": st["episodes"], "Training Iter": agent.training_iteration, "Epoch": 0,
                                 "In Heatup": int(self.phase == RunPhase.HEATUP),
                                 "ER #Transitions": agent.memory.num_transitions(),
                                 "Total steps": agent.total_steps_counter,
                                 "Training Reward": st["mean_return"], "Episode Length": st["mean_length"],
                                 "Loss/Mean": loss})
            return
        finished = agent.pop_finished_episodes()
        if not finished:
            return
        stats = agent.signal_stats.flush()
        eps = ""
        pol = getattr(agent, "exploration_policy", None)
        if pol is not None and hasattr(pol, "epsilon"):
            eps = pol.epsilon()
        elif pol is not None and hasattr(pol, "noise_schedule"):
            eps = pol.noise_schedule.current_value
        train = self.phase == RunPhase.TRAIN
        for env, length, ret in finished:
            self._episodes_logged += 1
            row = {"Episode #": self._episodes_logged, "Training Iter": agent.training_iteration, "Epoch": 0,
                   "In Heatup": int(self.phase == RunPhase.HEATUP),
                   "ER #Transitions": agent.memory.num_transitions(), "ER #Episodes": agent.memory.length(),
                   "Episode Length": length, "Total steps": agent.total_steps_counter, "Epsilon": eps,
                   "Shaped Training Reward": ret if train else float("nan"),
                   "Training Reward": ret if train else float("nan"),
                   "Update Target Network": int(getattr(agent, "_target_updated_since_log", False))}
            row.update(stats)
            stats = {k: "" for k in stats}                   # the statistics belong to the first row of the step
            self.logger.write(**row)
        agent._target_updated_since_log = False

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
                             "Epoch": 0, "In Heatup": 0, "Total steps": self.agent.total_steps_counter,
                             "Evaluation Reward": reward, "Shaped Evaluation Reward": reward})
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
