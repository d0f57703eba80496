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
from .graph_manager import ScheduleParameters  # noqa: F401  (presets import it from either module)


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
        self.use_graphs = True           # False: the agent issues every launch eagerly (same calls, same results)
        self.environment = self.agent = None
        self.logger = CsvLogger(csv_path)
        self.total_steps_counters = {RunPhase.HEATUP: 0, RunPhase.TRAIN: 0, RunPhase.TEST: 0}
        self.training_steps = 0
        self.phase = RunPhase.UNDEFINED
        self._episodes_logged = 0
        self._mid_episode = False        # some env is inside an episode that the next period must abandon

    # ------------------------------------------------------------------ graph creation (:62-78)
    def create_graph(self):
        rank = self.dist.rank if self.dist is not None else 0
        from ..compat import resolve_reference_style
        resolve_reference_style(self.agent_params, self.env_params)
        self.environment = dynamic_import(self.env_params.path)(self.env_params, self.device, rank=rank)
        self.agent = dynamic_import(self.agent_params.path)(self.agent_params, self.environment, self.device,
                                                            dist=self.dist, use_graphs=self.use_graphs)
        if self.visualization_parameters.dump_csv and hasattr(self.agent, "enable_signal_statistics"):
            self.agent.enable_signal_statistics()
            self.logger.set_signals(self.agent.SIGNAL_NAMES)
        return self

    def verify_graph_was_created(self):
        if self.agent is None:
            self.create_graph()

    def _set_phase(self, phase):
        """GraphManager.phase setter (graph_manager.py:333-344): the level managers AND the environments follow."""
        self.phase = phase
        self.agent.phase = phase
        self.environment.phase = phase

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

    def _episodes_ended_on_last_step(self):
        """how many envs finished an episode on the vector step just taken (each is one EnvironmentEpisodes unit)."""
        agent = self.agent
        if not getattr(agent, "_episode_just_ended", False):
            return 0
        ended = getattr(agent, "ended_episode_lengths", None)
        return len(ended) if ended is not None else agent.n_env

    def _fresh_episodes(self):
        """reset_internal_state(force_environment_reset=True) at the start of a period (graph_manager.py:411-424,
        477) unless every env stands at the first step of an episode already (the reset that closes an evaluation,
        or nothing played yet): the running episodes are abandoned, their last response is never observed."""
        if self._mid_episode and hasattr(self.agent, "reset_internal_state"):
            first = self.agent.reset_internal_state()
            if not getattr(self.agent, "restarts_memory_on_reset", False) and hasattr(self.agent.memory, "reset"):
                self.agent.memory.reset(first)           # (the on-policy agent restarts its rollout's frame stack itself)
        self._mid_episode = False

    def heatup(self, steps):                                                   # :400-424
        self.verify_graph_was_created()
        if steps.num_steps > 0:
            self._set_phase(RunPhase.HEATUP)
            done = 0
            while done < steps.num_steps:
                done += self._act()
            self._mid_episode = self._episodes_ended_on_last_step() != self.agent.n_env

    def train_and_act(self, steps):                                            # :463-489
        """act + train until `steps` have passed, in the unit `steps` is given in (current_step_counter[type(steps)],
        graph_manager.py:436,479): environment steps, finished episodes (each env's episode counts) or training
        iterations.  Off-policy agents that offer it take the one-record / one-graph step (DQN step_and_train) — the
        same call order, host draws and kernels as act() followed by train()."""
        self.verify_graph_was_created()
        if steps.num_steps <= 0:
            return
        self._set_phase(RunPhase.TRAIN)
        self._fresh_episodes()
        agent = self.agent
        fused = getattr(agent, "step_and_train", None)
        done = 0
        while done < steps.num_steps:
            before = agent.training_iteration
            if fused is not None and agent._step_graph_ok():
                fused()
                n = agent.n_env
                self.total_steps_counters[self.phase] += n
                self._log_finished_episodes()
            else:
                n = self._act()
                agent.train()
            trained = agent.training_iteration - before
            self.training_steps += trained
            done += n if isinstance(steps, EnvironmentSteps) else \
                self._episodes_ended_on_last_step() if isinstance(steps, EnvironmentEpisodes) else trained
        self._mid_episode = self._episodes_ended_on_last_step() != agent.n_env

    def evaluate(self, steps):                                                 # :491-523
        """Act greedily for `steps` episodes per env without storing or training; returns the mean
        evaluation reward (the success-rate early exit of the reference is not used here)."""
        self.verify_graph_was_created()
        if steps.num_steps <= 0 or not hasattr(self.agent, "evaluate_episodes"):
            return None
        self._set_phase(RunPhase.TEST)
        before = self.environment.total_steps
        episodes = steps.num_steps if isinstance(steps, EnvironmentEpisodes) else \
            max(1, -(-steps.num_steps // (self.agent.n_env * self.agent.L)))   # "at least `steps`", whole episodes
        reward = self.agent.evaluate_episodes(episodes)
        self._mid_episode = False                        # evaluate_episodes leaves every env freshly reset
        self.total_steps_counters[RunPhase.TEST] += self.environment.total_steps - before
        self.logger.write(**{"Episode #": self._episodes_logged, "Training Iter": self.agent.training_iteration,
                             "Epoch": 0, "In Heatup": 0, "Total steps": self.agent.total_steps_counter,
                             "Evaluation Reward": reward, "Shaped Evaluation Reward": reward})
        self._set_phase(RunPhase.TRAIN)
        return reward

    # ---------------------------------------------------------------- golden-threshold harness
    def validation_status(self, win_size=10):
        """What rl_coach/tests/test_golden.py:103-170 computes from the experiment CSV while a preset runs: the
        'Evaluation Reward' column without its empty cells, averaged over a window of `win_size` evaluations
        (np.convolve with ones(min(len, win)) / win, mode 'valid' — the reference divides by win_size even while
        fewer evaluations exist), compared with `preset_validation_params.min_reward_threshold`; the run has failed
        once 'Episode #' reaches `max_episodes_to_achieve_reward` without that."""
        import numpy as np
        pv = self.preset_validation_params
        rewards = np.array([float(r["Evaluation Reward"]) for r in self.logger.rows
                            if r.get("Evaluation Reward", "") != ""], dtype=np.float64)
        rewards = rewards[~np.isnan(rewards)]
        averaged = np.convolve(rewards, np.ones(min(len(rewards), win_size)) / win_size, mode='valid') \
            if len(rewards) >= 1 else np.array([0.0])
        episode = max([int(r["Episode #"]) for r in self.logger.rows if r.get("Episode #", "") != ""] or [0])
        passed = bool(np.any(averaged >= pv.min_reward_threshold))
        return {"passed": passed, "episode": episode, "averaged_rewards": averaged,
                "exhausted": episode >= pv.max_episodes_to_achieve_reward,
                "min_reward_threshold": pv.min_reward_threshold,
                "max_episodes_to_achieve_reward": pv.max_episodes_to_achieve_reward}

    def run_preset_validation(self, time_limit=60 * 60, win_size=10):
        """The golden test of a preset (test_golden.py:103-170) in-process: improve() until the averaged evaluation
        reward reaches the preset's threshold (pass) or the episode budget / time limit is spent (fail).  Returns the
        final validation_status() plus 'reason'."""
        if not self.preset_validation_params.test:
            raise ValueError("this preset declares no golden test (preset_validation_params.test is False)")
        t0 = time.time()

        def stop():
            st = self.validation_status(win_size)
            return st["passed"] or st["exhausted"] or time.time() - t0 > time_limit
        self.improve(should_stop=stop)
        st = self.validation_status(win_size)
        st["reason"] = "passed" if st["passed"] else \
            ("insufficient reward" if st["exhausted"] else "time limit" if time.time() - t0 > time_limit
             else "improve_steps exhausted")
        st["wall_s"] = time.time() - t0
        return st

    def improve(self, should_stop=None):                                       # :525-556
        """should_stop: optional callable checked after every evaluation period (the reference's evaluate() returns
        should_stop(), :519-523; the golden harness passes its own)."""
        self.verify_graph_was_created()
        self.heatup(self.schedule.heatup_steps)
        imp = self.schedule.improve_steps
        counter = (lambda: self.training_steps) if isinstance(imp, TrainingSteps) else \
            (lambda: self.total_steps_counters[RunPhase.TRAIN])
        count_end = counter() + imp.num_steps
        while counter() < count_end:
            self.train_and_act(self.schedule.steps_between_evaluation_periods)
            self.evaluate(self.schedule.evaluation_steps)
            if should_stop is not None and should_stop():
                break
        self.check_status()
        return self.logger.rows

    def check_status(self):
        """Device-side error bits of the agent (memory, networks) AND of the environment (an action outside the action
        space, dynamics outside a table's domain): raised here, once per improve(), not silently dropped."""
        for obj in (self.agent, self.environment):
            if hasattr(obj, "check_status"):
                obj.check_status()
