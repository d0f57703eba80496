"""Clipped PPO on one MI355X: vectorised rollout + GAE + clipped-surrogate updates, all in HBM.

Host-side mirror of rl_coach/agents/clipped_ppo_agent.py (parameter classes :41-131 keep the
reference's names, fields and defaults; ClippedPPOAgent keeps train / fill_advantages /
train_network / post_training_commands / choose_action) driving librlx kernels instead of
numpy + TF sessions.  What changes, and why it is still the same algorithm:

  * N envs step in lockstep (the reference has one env per process, level_manager.py:215-269);
    the training set is the concatenation of the N completed episodes, exactly what
    EpisodicExperienceReplay.transitions holds after N episodes.
  * the old-policy forward pass (target network) is evaluated once per training phase instead of
    once per minibatch (the reference's own TODO, clipped_ppo_agent.py:236-241): the target weights
    are frozen during train_network, so the values are identical.
  * host RNG draws (np.random for action sampling, random.shuffle for minibatch order) are made by
    the same generators in the same order as the reference and shipped to the device.
  * the per-step and per-minibatch launch sequences are launch-bound (tens of ~10 us kernels), so
    each is captured once into a hipGraph (torch.cuda.CUDAGraph on the current stream) and
    replayed; use_graphs=False runs the same calls eagerly.
"""
import os
import random

import numpy as np
import torch

from .. import _rlx
from ..core_types import EnvironmentSteps, RunPhase
from ..memories.episodic.episodic_rollout_buffer import DeviceEpisodicRolloutBuffer
from ..nn.networks import ClippedPPONet
from ..schedules import ConstantSchedule


from ..architectures.scheme_views import SchemeViews


class ClippedPPONetworkParameters(SchemeViews):               # clipped_ppo_agent.py:41-58
    def __init__(self):
        self.activation_function = 'tanh'
        self.embedder_scheme = 'Medium'
        self.middleware_scheme = 'Medium'
        self.batch_size = 64
        self.optimizer_type = 'Adam'
        self.learning_rate = 0.00025                     # base_parameters.py NetworkParameters default
        self.adam_optimizer_beta1 = 0.9
        self.adam_optimizer_beta2 = 0.99
        self.optimizer_epsilon = 0.0001
        self.clip_gradients = None
        self.use_separate_networks_per_head = True
        self.async_training = False
        self.l2_regularization = 0
        self.create_target_network = True
        self.shared_optimizer = True
        self.scale_down_gradients_by_number_of_workers_for_sync_training = True


class ClippedPPOAlgorithmParameters(object):             # clipped_ppo_agent.py:61-118
    def __init__(self):
        self.discount = 0.99
        self.num_episodes_in_experience_replay = 1000000
        self.gae_lambda = 0.95
        self.use_kl_regularization = False
        self.clip_likelihood_ratio_using_epsilon = 0.2
        self.estimate_state_value_using_gae = True
        self.beta_entropy = 0.01
        self.num_consecutive_playing_steps = EnvironmentSteps(2048)
        self.num_consecutive_training_steps = 1
        self.optimization_epochs = 10
        self.clipping_decay_schedule = ConstantSchedule(1)
        self.act_for_full_episodes = True
        self.reward_clipping = (-1.0, 1.0)               # Atari input filter (gym_environment.py:106-113)
        self.reward_rescale = 1.0
        # pre_network_filter = ObservationNormalizationFilter (presets/Mujoco_ClippedPPO.py): running
        # statistics updated on the whole rollout at train time (update_pre_network_filters_state_on_train)
        self.normalize_observations = False


class ClippedPPOAgentParameters(object):                 # clipped_ppo_agent.py:121-131
    def __init__(self):
        self.algorithm = ClippedPPOAlgorithmParameters()
        self.network_wrappers = {"main": ClippedPPONetworkParameters()}
        self.seed = 0

    @property
    def path(self):
        return 'coach_amd.agents.clipped_ppo_agent:ClippedPPOAgent'


def _capture(fn):
    from .vector_agent import capture
    return capture(fn)


class ClippedPPOAgent(object):
    epoch_graph = True      # one hipGraph per epoch, not per minibatch
    ragged = False          # envs end their episodes on different steps (set per instance from the env)
    _device_env = True
    restarts_memory_on_reset = True     # reset_internal_state() also restarts the rollout buffer's frame stack
    # an acting step's small launches merged: softmax + categorical draw (rlx_softmax_categorical_sample); reward filter +
    # episode totals + the action / reward / game_over columns (rlx_rollout_observe_step).  12 -> 9 launches per vector step
    # of the image agent; the flag is for same-process A/Bs and the tests that compare both forms
    FUSE_ACTING_LAUNCHES = True
    # discrete image agent: an acting step runs BOTH towers and leaves V(s) and the action probabilities in two rollout
    # columns (ClippedPPONet.act_and_record) — the weights do not change between a rollout and its training phase and the
    # old policy of that phase is the acting policy, so the V(s) pass over the whole dataset and the old-policy pass of
    # fill_advantages recompute exactly those numbers; with the columns recorded they are two gathers.  The value
    # tower rides in the same launches as the policy tower (the fused convolution launch has a workgroup per half image
    # and tower: 128 -> 256 of the chip's 256 CUs).  The flag is for same-process A/Bs and the equivalence test
    RECORD_WHILE_ACTING = True
    DATASET_CHUNK = 2048    # rows per forward pass of the whole-dataset passes (V(s) for GAE, the old policy): 256 -> 2048 is -1.0 ms per C2 iteration (profiles/r04_ab_ppo_chunk.txt)

    def __init__(self, agent_parameters, environment, device=None, dist=None, use_graphs=None):
        self.ap = agent_parameters
        self.env = environment
        self.device = device or environment.device
        self.dist = dist if (dist is not None and dist.enabled) else None
        self.lib = _rlx.lib()
        self.use_graphs = True if use_graphs is None else bool(use_graphs)
        # data parallel: reduce the FC + head gradients (95 % of the bytes, final before the convolution backward starts)
        # underneath the convolution backward?  None = decided from measurements at the first training phase
        # (_decide_overlap); True / False = forced (tests, A/B)
        self.overlap_allreduce = None if self.dist is not None else False
        self.overlap_decision = None
        alg, net = self.ap.algorithm, self.ap.network_wrappers["main"]
        ep = environment.p
        self.continuous = ep.action_dim is not None      # BoxActionSpace -> continuous PPO head
        self.n_env, self.A = ep.num_envs, (ep.action_dim if self.continuous else ep.num_actions)
        self.image = ep.kind == "image"
        self.stack = 4
        self.L = ep.episode_length
        obs_shape = tuple(ep.observation_shape) + (self.stack,) if self.image else tuple(ep.observation_shape)
        # seeds: host generators exactly like Agent.__init__ (agents/agent.py:49-55); every rank
        # builds identical initial weights, env streams differ through env_id0
        if self.ap.seed is not None:
            random.seed(self.ap.seed)
            np.random.seed(self.ap.seed)
        self.networks = {"main": ClippedPPONet(
            self.device, obs_shape, self.A, activation=net.activation_function,
            embedder=net.embedder_scheme, middleware=net.middleware_scheme,
            learning_rate=net.learning_rate, adam_beta1=net.adam_optimizer_beta1,
            adam_beta2=net.adam_optimizer_beta2, optimizer_epsilon=net.optimizer_epsilon,
            clip_likelihood_ratio_using_epsilon=alg.clip_likelihood_ratio_using_epsilon,
            beta_entropy=alg.beta_entropy, seed=self.ap.seed or 0, continuous=self.continuous)}
        if self.dist is not None and self.ap.seed is not None:
            # rank-offset sampling seed (coach.py:746 seed + task_index) AFTER the shared weight init
            random.seed(self.ap.seed + self.dist.rank)
            np.random.seed(self.ap.seed + self.dist.rank)
        # act_for_full_episodes: train once >= num_consecutive_playing_steps were played AND every
        # env's episode is complete (agents/agent.py:681-699); with fixed-length episodes that is a
        # whole number of episodes per env.
        per_env = -(-alg.num_consecutive_playing_steps.num_steps // self.n_env)
        self.steps_per_phase = -(-per_env // self.L) * self.L
        # Envs that end their episodes on DIFFERENT steps (per-env time limits, emulators): the rule above becomes
        # "train once the complete episodes hold num_consecutive_playing_steps transitions" (for one env: the
        # reference's rule), which takes at most ceil(steps / n_env) + longest episode - 1 vector steps.
        self._device_env = hasattr(environment, "launch_step") and hasattr(environment, "host_tick")
        lens = getattr(environment, "lengths_host", None)
        self.ragged = getattr(environment, "dones_host", None) is not None and \
            (lens is None or getattr(environment, "lengths", None) is not None)     # per-env limits were declared
        if self.ragged:
            longest = int(max(lens)) if lens is not None else int(self.L)
            # + the episode that overshoots the phase; envs whose episode length is DATA (no declared lengths) may also
            # lose open episodes to forced resets between two training phases (evaluation periods): room for three
            self.steps_per_phase = per_env + longest * (1 if lens is not None else 4)
        self.memory = DeviceEpisodicRolloutBuffer(
            self.device, self.n_env, self.steps_per_phase,
            frame_shape=ep.observation_shape if self.image else None, stack=self.stack,
            obs_dim=None if self.image else int(ep.observation_shape[0]),
            action_dim=self.A if self.continuous else None)
        self._rec_missing = False       # a stored step of the current rollout was NOT recorded (another phase / path)
        dev, n = self.device, self.n_env
        self.actions = torch.zeros((n, self.A), dtype=torch.float32, device=dev) if self.continuous else \
            torch.zeros(n, dtype=torch.int32, device=dev)
        self.norm = None
        if alg.normalize_observations:
            if self.image:
                raise ValueError("ObservationNormalizationFilter applies to vector observations")
            from ..filters.observation import ObservationNormalizationFilter
            self.norm = ObservationNormalizationFilter(int(ep.observation_shape[0]), dev)
            self.act_obs = torch.empty(n, int(ep.observation_shape[0]), dtype=torch.float32, device=dev)
        self._records_acting()          # allocates the rollout's V(s) / probability columns where they are used
        if self.continuous:
            lo = np.broadcast_to(np.asarray(ep.action_low, dtype=np.float32), (self.A,)).copy()
            hi = np.broadcast_to(np.asarray(ep.action_high, dtype=np.float32), (self.A,)).copy()
            self.d_low, self.d_high = torch.from_numpy(lo).to(dev), torch.from_numpy(hi).to(dev)
        self.filtered_reward = torch.zeros(n, dtype=torch.float32, device=dev)
        self.ep_return = torch.zeros(n, dtype=torch.float64, device=dev)
        self.ep_len = torch.zeros(n, dtype=torch.int32, device=dev)
        self.ep_acc = torch.zeros(8, dtype=torch.float64, device=dev)
        self.lib.episode_stats_init(self.ep_return, self.ep_len, n, self.ep_acc, _rlx.current_stream())
        self.phase = RunPhase.TRAIN
        self.total_steps_counter = 0
        self.last_training_phase_step = 0
        self.training_iteration = 0
        self.signals = {}
        self._graphs = {}
        self._warm = set()
        self._episode_steps = np.zeros(self.n_env, dtype=np.int64)
        self._episode_just_ended = False
        self._draw_pool, self._draw_pos, self._draw_table_from = None, 0, None
        self._eval_mem = None
        self._alloc_training_buffers()
        self.memory.reset(self.env.reset_internal_state())

    # ------------------------------------------------------------------------------- buffers
    def _alloc_training_buffers(self):
        from ..staging import Stager
        dev, B = self.device, self.ap.network_wrappers["main"].batch_size
        cap = self.memory.cap
        f32, f64 = torch.float32, torch.float64
        self.ds_reward = torch.empty(cap, dtype=f32, device=dev)
        self.ds_done = torch.empty(cap, dtype=torch.uint8, device=dev)
        act_shape = (cap, self.A) if self.continuous else (cap,)
        self.ds_action = torch.empty(act_shape, dtype=self.actions.dtype, device=dev)
        self.ds_value = torch.empty(cap, dtype=f32, device=dev)
        self.ds_adv64 = torch.empty(cap, dtype=f64, device=dev)
        self.ds_adv = torch.empty(cap, dtype=f32, device=dev)
        self.ds_vtarget = torch.empty(cap, dtype=f32, device=dev)
        self.ds_old_probs = torch.empty(cap, self.A, dtype=f32, device=dev)     # discrete: probs; continuous: mean
        self.ds_old_std = torch.empty(cap, self.A, dtype=f32, device=dev) if self.continuous else None
        if self.norm is not None:
            D = self.memory.obs_dim
            self.ds_obs_raw = torch.empty(cap, D, dtype=f32, device=dev)
            self.ds_obs = torch.empty(cap, D, dtype=f32, device=dev)
        self.adv_stats = torch.empty(2, dtype=f64, device=dev)
        obs_tail = tuple(self.memory.cur_state.shape[1:])
        odt = self.memory.cur_state.dtype
        self.chunk = min(int(self.DATASET_CHUNK), cap)
        self.chunk_obs = torch.empty((self.chunk,) + obs_tail, dtype=odt, device=dev)
        self.mb_obs = torch.empty((B,) + obs_tail, dtype=odt, device=dev)
        self.mb_rows = torch.empty(B, dtype=torch.int32, device=dev)
        # the epoch's shuffled dataset order lives in ONE static device buffer; minibatch i of the
        # epoch reads its slice in place (its hipGraph is keyed by i), so there is no per-minibatch copy
        self._perm = Stager((cap,), torch.int32, dev, depth=4)
        self.perm_dev = self._perm.dst
        self.mb_action = torch.empty((B, self.A) if self.continuous else (B,), dtype=self.actions.dtype, device=dev)
        self.mb_old_std = torch.empty(B, self.A, dtype=f32, device=dev) if self.continuous else None
        self.mb_adv = torch.empty(B, dtype=f32, device=dev)
        self.mb_vtarget = torch.empty(B, dtype=f32, device=dev)
        self.mb_old = torch.empty(B, self.A, dtype=f32, device=dev)
        self.mb_ratio = torch.empty(B, dtype=f32, device=dev)
        self.mb_clipped = torch.empty(B, dtype=f32, device=dev)
        self.scalar_acc = torch.zeros(8, dtype=f32, device=dev)
        # one uniform per (step, env): what the phase's np.random.choice calls consume (continuous:
        # A standard normals per (step, env) behind np.random.normal(mean, std), additive_noise.py:106)
        shape = (self.steps_per_phase, self.n_env, self.A) if self.continuous else (self.steps_per_phase, self.n_env)
        if self.ragged:
            # a phase's length is not known in advance: the draws of ONE step are shipped per step, so that the host
            # stream is consumed exactly as n_env sequential np.random calls per step would
            shape = (1,) + shape[1:]
        self._uniforms = Stager(shape, f64, dev, depth=32 if self.ragged else 4)
        self.uniforms_all = self._uniforms.dst

    # ------------------------------------------------------------------------------ graph util
    def _run(self, key, fn):
        """Run `fn` (pure device work on static buffers): eagerly the first time (allocates cached
        buffers / offset tables), captured into a hipGraph the second time, replayed afterwards."""
        if not self.use_graphs:
            return fn()
        g = self._graphs.get(key)
        if g is not None:
            return g.replay()
        if key not in self._warm:
            self._warm.add(key)
            return fn()
        torch.cuda.synchronize()
        self._graphs[key] = _capture(fn)
        return self._graphs[key].replay()

    def _refill_draws(self, from_step):
        """Lockstep envs: the host draws of the rest of the phase — one uniform (A normals) per (step, env), in the
        order n_env sequential np.random.choice / np.random.normal calls per step make them (nothing else consumes
        np.random while acting) — shipped as ONE table indexed by the step.  Draws that were made for steps which
        never ran (a phase that trained early, the steps a forced reset abandoned) are the next ones of the stream:
        they are used before anything new is drawn, so the stream is consumed exactly as step-by-step draws would."""
        shape = (self.n_env, self.A) if self.continuous else (self.n_env,)
        need = self.steps_per_phase - from_step
        left = self._draw_pool[self._draw_pos:] if self._draw_pool is not None else np.zeros((0,) + shape)
        if left.shape[0] < need:
            m = need - left.shape[0]
            fresh = np.random.standard_normal((m,) + shape) if self.continuous else np.random.random_sample((m,) + shape)
            left = np.concatenate([left, fresh], 0)
        self._draw_pool, self._draw_pos = left, 0
        table = np.zeros((self.steps_per_phase,) + shape, dtype=np.float64)
        table[from_step:] = left[:need]
        self._uniforms.push(table)
        self._draw_table_from = from_step

    # --------------------------------------------------------------------------------- acting
    def _act_device(self, step):
        """Device work of one vector step: stacked states -> online policy tower -> categorical
        sample -> env step -> reward filter -> episode stats -> store."""
        s = _rlx.current_stream()
        alg = self.ap.algorithm
        states = self.memory.current_states()
        if self.norm is not None:          # run_pre_network_filter_for_inference (:346-350): no state update
            states = self.norm.filter(states, update_internal_state=False, out=self.act_obs)
        if self.continuous:
            mean, std = self.networks["main"].policy_mean_std(states, self.n_env)
            if self.phase == RunPhase.TRAIN:                                # additive_noise.py:99-106
                # (not clipped: the transition keeps the sampled action, agent.py:935 — the likelihood ratio is taken at it —
                # and the environment clips what it executes, environments/environment.py:283)
                self.lib.gaussian_action(mean, None, std, self.uniforms_all[0 if self.ragged else step], None, None,
                                         self.n_env, self.A, self.actions, s)
            else:
                self.actions.copy_(mean)
        else:
            if self.phase == RunPhase.TRAIN and self._records_acting():
                m, r0 = self.memory, step * self.n_env
                self.networks["main"].act_and_record(states, self.n_env, self.uniforms_all[0 if self.ragged else step],
                                                     self.actions, m.act_value[r0:r0 + self.n_env],
                                                     m.act_probs[r0:r0 + self.n_env])
            elif self.phase == RunPhase.TRAIN and self.FUSE_ACTING_LAUNCHES:
                # softmax + categorical draw (categorical.py:45-48) as one launch
                self.networks["main"].policy_probs(states, self.n_env,
                                                   sample=(self.uniforms_all[0 if self.ragged else step], self.actions))
            else:
                probs = self.networks["main"].policy_probs(states, self.n_env)
                if self.phase == RunPhase.TRAIN:
                    self.lib.categorical_sample(probs, self.A, self.uniforms_all[0 if self.ragged else step], self.n_env,
                                                self.A, self.actions, s)        # categorical.py:45-48
                else:
                    self.lib.argmax_rows(probs, self.A, self.n_env, self.A, self.actions, s)     # :50-56
        if self._device_env:
            # device-resident env: the launch is part of the (captured) step, its host clock ticks in act()
            self.env.launch_step()
            env = self.env
            next_obs, reset_obs, reward, game_over = env.next_obs, env.reset_obs, env.reward, env.game_over
        else:
            next_obs, reset_obs, reward, game_over = self.env.step(self.actions)
        has_clip = alg.reward_clipping is not None
        lo, hi = alg.reward_clipping if has_clip else (0.0, 0.0)
        if self.FUSE_ACTING_LAUNCHES and self.n_env <= 1024:
            # reward filter -> episode totals -> the step's action / reward / game_over columns: one launch
            m = self.memory
            row0 = step * self.n_env
            self.lib.rollout_observe_step(reward, self.filtered_reward, float(alg.reward_rescale), int(has_clip), float(lo),
                                          float(hi), game_over, self.ep_return, self.ep_len, self.ep_acc, None, None,
                                          self.actions, self.actions.element_size() * (self.actions.numel() // self.n_env),
                                          m.action, m.reward, m.game_over, row0, m.cap, self.n_env, m.status, s)
            m.store_step_at(step, self.actions, self.filtered_reward, game_over, next_obs, reset_obs, columns=False)
            return
        self.lib.reward_filter(reward, self.filtered_reward, self.n_env, alg.reward_rescale,
                               int(has_clip), lo, hi, s)
        self.lib.episode_stats_step(self.filtered_reward, game_over, self.ep_return, self.ep_len,
                                    self.n_env, self.ep_acc, None, None, s)
        self.memory.store_step_at(step, self.actions, self.filtered_reward, game_over, next_obs, reset_obs)

    def _records_acting(self):
        """V(s) and the action probabilities are recorded by the acting step (RECORD_WHILE_ACTING): the discrete image
        agent without an observation-normalisation filter (that filter's statistics move between acting and training,
        clipped_ppo_agent.py:320-322, so its V(s) pass is NOT a recomputation)."""
        rec = self.RECORD_WHILE_ACTING and self.image and not self.continuous and self.norm is None
        if rec and not hasattr(self.memory, "act_value"):
            self.memory.enable_policy_columns(self.A)
        return rec

    def act(self):
        """One vector step: observe -> act -> env.step (LevelManager.step, level_manager.py:215-269)."""
        step = self.memory.steps
        if step >= self.memory.T:
            raise ValueError("rollout buffer is full; call train()")
        if not (self.phase == RunPhase.TRAIN and self._records_acting()):
            self._rec_missing = True            # this step's rows carry no recorded V(s) / probabilities
        if self.ragged and self.phase == RunPhase.TRAIN:
            self._uniforms.push(np.random.standard_normal((1, self.n_env, self.A)) if self.continuous
                                else np.random.random_sample((1, self.n_env)))
        elif self.phase == RunPhase.TRAIN:
            if step == 0 or self._draw_table_from is None:
                self._refill_draws(step)
            self._draw_pos += 1
        self.ap.algorithm.clipping_decay_schedule.step()                    # choose_action (:352-354)
        if self._device_env:
            self._run(("act", step, self.phase), lambda: self._act_device(step))
            self.env.host_tick()
        else:
            self._act_device(step)            # CPU emulators behind the env: not a pure device step, never captured
        self.memory.steps += 1
        if self.ragged:
            dones = self.env.dones_host
            self.memory.note_episode_ends(dones)
            self._episode_steps += 1
            ended = np.nonzero(dones)[0]
            self._episode_just_ended = ended.size > 0
            if ended.size:
                self.ended_episode_lengths = self._episode_steps[ended].copy()
                self._episode_steps[ended] = 0
        else:
            self._episode_just_ended = self.memory.steps % self.L == 0
            self.ended_episode_lengths = np.full(self.n_env, self.L, dtype=np.int64)
        self.env.total_steps += self.n_env
        self.total_steps_counter += self.n_env
        return self.n_env

    # ----------------------------------------------------------------------- resets / evaluation
    def reset_internal_state(self):
        """GraphManager.reset_internal_state(force_environment_reset=True) + Agent.reset_internal_state
        (graph_manager.py:411-424, agent.py:603-629): every env starts a new episode now; the running episodes were
        held in current_episode_buffer and never reach the memory (lockstep envs: their rows and ring frames are
        rewound, the next episode overwrites them; the host draws made for the abandoned steps' successors are the
        next ones of the np.random stream and are used first)."""
        self.memory.drop_open_episodes(None if self.ragged else self.L)
        self._draw_table_from = None                     # the step -> draw table no longer matches the stream position
        self._episode_steps[:] = 0
        self._episode_just_ended = False
        self.lib.episode_stats_init(self.ep_return, self.ep_len, self.n_env, None, _rlx.current_stream())
        first = self.env.reset_internal_state()
        self.memory.reset(first)
        return first

    def _evaluation_memory(self):
        """Scratch frame stack of the evaluation episodes (image observations): a one-step rollout buffer whose ring
        holds stack + 6 frames per env — the training rollout's ring, rows and cursors are never written in TEST
        phase (Agent.observe_transition stores only in TRAIN / HEATUP, agent.py:956-962)."""
        if self._eval_mem is None:
            self._eval_mem = DeviceEpisodicRolloutBuffer(self.device, self.n_env, 1, frame_shape=self.env.p.observation_shape,
                                                         stack=self.stack)
        return self._eval_mem

    def _evaluate_device(self, scratch):
        """Device work of one evaluation step (hipGraph-capturable): state -> most probable action / policy mean ->
        env step -> scratch frame stack.  Nothing of the training rollout is read or written."""
        s, n = _rlx.current_stream(), self.n_env
        states = scratch.current_states() if scratch is not None else self._eval_state
        if self.norm is not None:
            states = self.norm.filter(states, update_internal_state=False, out=self.act_obs)
        if self.continuous:
            mean, _ = self.networks["main"].policy_mean_std(states, n)
            self.actions.copy_(mean)
        else:
            probs = self.networks["main"].policy_probs(states, n)
            self.lib.argmax_rows(probs, self.A, n, self.A, self.actions, s)
        if self._device_env:
            self.env.launch_step()
            env = self.env
            next_obs, reset_obs, reward, game_over = env.next_obs, env.reset_obs, env.reward, env.game_over
        else:
            next_obs, reset_obs, reward, game_over = self.env.step(self.actions)
        self._eval_total.add_(reward.double() * self._eval_active)
        if scratch is not None:
            scratch.store_step_at(0, self.actions, reward, game_over, next_obs, reset_obs)
        else:
            self._eval_state.copy_(torch.where(game_over.view(-1, 1).bool(), reset_obs, next_obs))

    def evaluate_episodes(self, episodes_per_env=1):
        """GraphManager.evaluate (graph_manager.py:491-523) for the env vector: every env is reset (the open episodes
        of the rollout are abandoned like at any forced reset), whole episodes are played in TEST phase — the most
        probable action (categorical.py:50-56) / the policy mean (additive_noise.py:99-106), observations through the
        pre-network filter WITHOUT updating its statistics (clipped_ppo_agent.py:346-350) — nothing is stored, no
        counter of the training schedule moves; training resumes from a fresh reset.  Image observations are stacked
        in a scratch ring (_evaluation_memory).  Returns the mean undiscounted episode reward over envs."""
        prev, prev_env = self.phase, getattr(self.env, "phase", None)
        self.phase = self.env.phase = RunPhase.TEST
        s = _rlx.current_stream()
        n, dev = self.n_env, self.device
        self.memory.drop_open_episodes(None if self.ragged else self.L)
        self._draw_table_from = None
        first = self.env.reset_internal_state()
        scratch = None
        if self.image:
            scratch = self._evaluation_memory()
            scratch.reset(first)
        else:
            if getattr(self, "_eval_state", None) is None:
                self._eval_state = torch.empty_like(self.memory.cur_state)
            self._eval_state.copy_(first)
        if getattr(self, "_eval_total", None) is None:
            self._eval_total = torch.zeros(n, dtype=torch.float64, device=dev)
            self._eval_active = torch.ones(n, dtype=torch.float64, device=dev)
        self._eval_total.zero_()
        self._eval_active.fill_(1.0)
        finished = np.zeros(n, dtype=np.int64)
        lockstep_t = 0
        try:
            while (finished < episodes_per_env).any():
                if self._device_env:
                    self._run(("evaluate",), lambda: self._evaluate_device(scratch))
                    self.env.host_tick()
                else:
                    self._evaluate_device(scratch)
                self.env.total_steps += n
                dh = getattr(self.env, "dones_host", None)
                if dh is None:                                   # envs without a host clock: all end after L steps
                    lockstep_t += 1
                    dh = np.full(n, lockstep_t % self.L == 0)
                was_active = finished < episodes_per_env
                finished += dh.astype(np.int64)
                active = finished < episodes_per_env
                if (active != was_active).any():                 # an env used up its quota: its further rewards do not count
                    self._eval_active.copy_(torch.from_numpy(active.astype(np.float64)))
        finally:
            self.phase = prev
            self.env.phase = prev_env if prev_env is not None else prev
            self._episode_steps[:] = 0
            self._episode_just_ended = False
            self.lib.episode_stats_init(self.ep_return, self.ep_len, n, None, s)
            self.memory.reset(self.env.reset_internal_state())
        return float(self._eval_total.mean().item()) / episodes_per_env

    # ------------------------------------------------------------------------------- training
    def _should_train(self):
        """agents/agent.py:662-699 with act_for_full_episodes: enough steps AND episodes complete."""
        steps = self.ap.algorithm.num_consecutive_playing_steps.num_steps
        if self.ragged:
            done = self.memory.num_transitions_in_complete_episodes()
            if done >= steps or (self.memory.steps >= self.memory.T and done > 0):
                if done < steps and self.dist is not None:
                    # every rank must run the same number of minibatch collectives per phase
                    raise ValueError("rollout buffer full with %d < %d complete-episode transitions: with data "
                                     "parallelism every rank trains on num_consecutive_playing_steps transitions; "
                                     "declare a longer max episode length" % (done, steps))
                self.last_training_phase_step = self.total_steps_counter
                return True
            if self.memory.steps >= self.memory.T:
                raise ValueError("no episode completed within %d vector steps: an episode is longer than the "
                                 "environment's declared maximum" % self.memory.T)
            return False
        enough = (self.total_steps_counter - self.last_training_phase_step) >= steps
        complete = self.memory.steps > 0 and self.memory.steps % self.L == 0
        if enough and complete and self.memory.num_transitions() > 0:
            self.last_training_phase_step = self.total_steps_counter
            return True
        return False

    def _fill_advantages_device(self, n, rows, n_train=None, recorded=False):
        """n_train: the transitions the phase trains on (dataset[:num_steps], :330-331) — the only ones whose old-policy
        outputs are read.  With episodes longer than the playing phase (L = 1024, 64 envs: a dataset of 65 536
        transitions, 2 048 trained on) the old-policy pass over the whole dataset would be 32 x the work the
        reference's per-minibatch pass does."""
        mem, net, alg = self.memory, self.networks["main"], self.ap.algorithm
        n_train = n if n_train is None else min(n, n_train)
        s = _rlx.current_stream()
        cols = [(mem.reward, self.ds_reward), (mem.game_over, self.ds_done), (mem.action, self.ds_action)]
        if recorded:       # V(s) and the old policy's probabilities were left in the rollout by the acting steps
            cols += [(mem.act_value, self.ds_value), (mem.act_probs, self.ds_old_probs)]
        mem.gather_columns(rows, n, cols)
        if self.norm is not None:
            # pre_network_filter over the whole dataset, statistics updated first (:320-322)
            mem.gather_states(rows, n, self.ds_obs_raw[:n])
            self.norm.filter(self.ds_obs_raw[:n], update_internal_state=self.dist is None, out=self.ds_obs[:n])
            if self.dist is None:
                # ... and InputFilter.filter then walks the NEXT states of the same Transitions (filters/filter.py:
                # 314-333): they enter the statistics after the states were normalised (their normalised values are
                # read by nobody).  Pinned by tests/golden/ppoc_loop.npz: the count grows by 2 n per phase
                mem.gather_next_states(rows, n, self.ds_obs_raw[:n])
                self.norm.push(self.ds_obs_raw[:n])

        def chunk_obs(c0, m):
            if self.norm is not None:
                return self.ds_obs[c0:c0 + m]
            return mem.gather_states(rows[c0:c0 + m], m, self.chunk_obs[:m])
        for c0 in range(0, 0 if recorded else n, self.chunk):
            m = min(self.chunk, n - c0)
            net.values(chunk_obs(c0, m), m, out=self.ds_value[c0:c0 + m])
        # lockstep: n_env sequences of equal length; ragged: the complete episodes back to back — ONE sequence, the scan
        # restarts at every game_over and every listed episode ends with one
        n_seq = 1 if self.ragged else mem.n_env
        self.lib.gae(self.ds_reward, self.ds_value, self.ds_done, None, n_seq, n // n_seq,
                     alg.discount, alg.gae_lambda, self.ds_adv64, self.ds_vtarget, s)
        self.lib.standardize(self.ds_adv64, n, self.ds_adv, None, self.adv_stats, s)
        # old policy = target network, frozen for the whole phase (:238-241, hoisted out of the loop)
        for c0 in range(0, 0 if recorded else n_train, self.chunk):
            m = min(self.chunk, n_train - c0)
            if self.continuous:
                net.policy_mean_std(chunk_obs(c0, m), m, use_target=True, tag="old",
                                    out_mean=self.ds_old_probs[c0:c0 + m], out_std=self.ds_old_std[c0:c0 + m])
            else:
                net.policy_probs(chunk_obs(c0, m), m, use_target=True, tag="old",
                                 out=self.ds_old_probs[c0:c0 + m])

    def fill_advantages(self):
        """clipped_ppo_agent.py:157-207 on device: V(s) for the whole dataset in chunks, segmented
        GAE scan per episode, (adv - mean) / std; plus the old-policy probabilities."""
        n = self.memory.num_transitions()
        rows = self.memory.dataset_rows()
        if self.norm is not None and self.dist is not None:
            # shared running statistics: the collective stays outside the captured graph
            self.memory.gather_states(rows, n, self.ds_obs_raw[:n])
            self.norm.push_shared(self.ds_obs_raw[:n], self.dist)
        n_train = min(n, self.ap.algorithm.num_consecutive_playing_steps.num_steps)
        # every stored step of this rollout left its V(s) / action probabilities behind (RECORD_WHILE_ACTING)?
        recorded = self._records_acting() and not self._rec_missing
        self._run(("fill", n, n_train, recorded), lambda: self._fill_advantages_device(n, rows, n_train, recorded))
        if self.norm is not None and self.dist is not None:
            # the next states of the dataset, after the states were normalised (see _fill_advantages_device)
            self.memory.gather_next_states(rows, n, self.ds_obs_raw[:n])
            self.norm.push_shared(self.ds_obs_raw[:n], self.dist)

    def _gather_minibatch(self, m, i=0):
        mem, s = self.memory, _rlx.current_stream()
        rows_all = mem.dataset_rows()
        B = self.ap.network_wrappers["main"].batch_size
        idx, rows = self.perm_dev[i * B:i * B + m], self.mb_rows[:m]
        # one launch gathers the per-transition columns AND translates dataset index -> storage row
        # (rows_all is one more 4-byte column over the same index)
        cols = [(rows_all, rows), (self.ds_action, self.mb_action), (self.ds_adv, self.mb_adv), (self.ds_vtarget, self.mb_vtarget),
                (self.ds_old_probs, self.mb_old)]
        if self.continuous:
            cols.append((self.ds_old_std, self.mb_old_std))
        if self.norm is not None:
            cols.append((self.ds_obs, self.mb_obs))           # normalised observations, dataset order
        self.lib.copy_columns(_rlx.make_columns(cols), len(cols), idx, None, 0, 0, self.ds_adv.numel(), m, m,
                              mem.status, s)
        if self.norm is not None:
            return self.mb_obs[:m]
        return mem.gather_states(rows, m, self.mb_obs[:m])

    def _gather_epoch(self, n):
        """The whole epoch's training set in minibatch order with TWO launches (the epoch's permutation is in perm_dev
        before its first minibatch runs): minibatch i then reads rows [i*B, i*B + m) of these buffers in place — 2
        launches per epoch instead of 2 per minibatch (a launch costs ~4.6 us here, as much as the gather itself)."""
        mem, s = self.memory, _rlx.current_stream()
        e = self._epoch_buffers()
        cols = [(mem.dataset_rows(), e["rows"]), (self.ds_action, e["action"]), (self.ds_adv, e["adv"]),
                (self.ds_vtarget, e["vtarget"]), (self.ds_old_probs, e["old"])]
        if self.continuous:
            cols.append((self.ds_old_std, e["old_std"]))
        if self.norm is not None:
            cols.append((self.ds_obs, e["obs"]))
        self.lib.copy_columns(_rlx.make_columns(cols), len(cols), self.perm_dev[:n], None, 0, 0, self.ds_adv.numel(), n, n,
                              mem.status, s)
        if self.norm is None:
            mem.gather_states(e["rows"][:n], n, e["obs"][:n])
        return e

    def _epoch_buffers(self):
        e = getattr(self, "_epoch_bufs", None)
        if e is None:
            cap, dev, f32 = self.memory.cap, self.device, torch.float32
            e = self._epoch_bufs = dict(
                rows=torch.empty(cap, dtype=torch.int32, device=dev),
                action=torch.empty((cap, self.A) if self.continuous else (cap,), dtype=self.actions.dtype, device=dev),
                adv=torch.empty(cap, dtype=f32, device=dev), vtarget=torch.empty(cap, dtype=f32, device=dev),
                old=torch.empty(cap, self.A, dtype=f32, device=dev),
                old_std=torch.empty(cap, self.A, dtype=f32, device=dev) if self.continuous else None,
                obs=torch.empty((cap,) + tuple(self.mb_obs.shape[1:]), dtype=self.mb_obs.dtype, device=dev))
        return e

    def _minibatch_fb(self, m, clip_rescaler, stop_after_dense=False, i=0, epoch=None):
        """epoch: the buffers of _gather_epoch — minibatch i's rows are already in training order."""
        if epoch is not None:
            B = self.ap.network_wrappers["main"].batch_size
            sl = slice(i * B, i * B + m)
            old = (epoch["old"][sl], epoch["old_std"][sl]) if self.continuous else epoch["old"][sl]
            self.networks["main"].forward_backward(epoch["obs"][sl], m, epoch["action"][sl], epoch["adv"][sl],
                                                   epoch["vtarget"][sl], old, clip_rescaler, self.mb_ratio,
                                                   self.mb_clipped, stop_after_dense=stop_after_dense)
            return
        obs = self._gather_minibatch(m, i)
        old = (self.mb_old, self.mb_old_std) if self.continuous else self.mb_old
        self.networks["main"].forward_backward(obs, m, self.mb_action, self.mb_adv, self.mb_vtarget,
                                               old, clip_rescaler, self.mb_ratio, self.mb_clipped,
                                               stop_after_dense=stop_after_dense)

    def _minibatch_finish(self, scale):
        net = self.networks["main"]
        net.finish_update(scale, signal_acc=self.scalar_acc)

    OVERLAP_EDGE_PAIR_US = 50.0     # what a fork / join pair of cross-stream edges per minibatch costs (measured
                                    # +50 us at world size 1 in round 1; every round-4 branch experiment lost for the same reason)

    def _decide_overlap(self, n, clip):
        """Two-bucket overlap or one blocking all-reduce per minibatch — chosen once, from this job's own numbers:
        the overlap can hide at most min(all-reduce of the late bucket, convolution backward) per minibatch and costs a
        cross-stream edge pair.  Both are MEASURED here (the collective alone, RCCL only; the convolution backward
        between two events on a dry forward / backward of the first minibatch — it writes activations and gradients
        that the real pass overwrites, no weight, Adam slot or signal sum changes).  Every rank takes the decision of
        the slowest rank.  Without graph-resident collectives (gloo, or a failed RCCL probe) the answer is the single
        blocking all-reduce: three graph segments and two eager collectives per minibatch measured as a net loss."""
        net, d = self.networks["main"], self.dist
        off = net.late_gradient_offset()
        rec = {"graph_resident": bool(self.use_graphs and d.capturable()), "edge_pair_us": self.OVERLAP_EDGE_PAIR_US,
               "late_bucket_allreduce_us": None, "conv_backward_us": None}
        if not rec["graph_resident"] or off <= 0:
            rec.update(overlap=False, why="collectives are not graph nodes on this backend" if off > 0
                       else "no early-finished bucket (no convolution layers)")
        else:
            B = self.ap.network_wrappers["main"].batch_size
            m = min(B, n)
            e = self._gather_epoch(n)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            us = []
            for _ in range(3):
                self._minibatch_fb(m, clip, True, i=0, epoch=e)
                ev[0].record()
                net.backward_rest()
                ev[1].record()
                ev[1].synchronize()
                us.append(1e3 * ev[0].elapsed_time(ev[1]))
            bw = d.max_over_ranks(min(us))
            ar = d.max_over_ranks(d.all_reduce_us(net.params.grads.numel() - off) or 0.0)
            hidden = min(ar, bw)
            rec.update(late_bucket_allreduce_us=round(ar, 1), conv_backward_us=round(bw, 1), overlap=bool(hidden > self.OVERLAP_EDGE_PAIR_US),
                       why="min(all-reduce, convolution backward) = %.1f us %s the %.0f us edge pair"
                           % (hidden, ">" if hidden > self.OVERLAP_EDGE_PAIR_US else "<=", self.OVERLAP_EDGE_PAIR_US))
        self.overlap_decision = rec
        self.overlap_allreduce = rec["overlap"]

    def train_network(self, order, epochs):
        """clipped_ppo_agent.py:209-308.  `order`: dataset indices after the reference's
        `dataset[:num_steps]; shuffle(dataset)`; every epoch re-shuffles (Batch.shuffle,
        core_types.py:452-468).  Returns the per-epoch means of
        [surrogate, entropy, kl, policy-head total, value loss, grad norm]."""
        alg = self.ap.algorithm
        netp = self.ap.network_wrappers["main"]
        B = netp.batch_size
        n = len(order)
        # clip_param_rescaler (:266-268) lives in a device scalar: the captured graphs below do not depend on its value
        # (a decaying clipping_decay_schedule, presets/Mujoco_ClippedPPO.py, would otherwise re-capture every phase)
        self.networks["main"].set_clip_rescaler(float(alg.clipping_decay_schedule.current_value))
        clip = None
        scale = self.dist.grad_scale(netp.scale_down_gradients_by_number_of_workers_for_sync_training) \
            if self.dist else 1.0
        results = []
        for j in range(epochs):
            batch_order = list(range(n))
            random.shuffle(batch_order)                                   # Batch.shuffle
            order = [order[i] for i in batch_order]
            full = np.zeros(self.perm_dev.numel(), dtype=np.int32)
            full[:n] = order
            self._perm.push(full)
            self.scalar_acc.zero_()
            nmb = -(-n // B)                                              # math.ceil (:232)
            if self.overlap_allreduce is None:
                self._decide_overlap(n, clip)
            in_graph = self.dist is not None and self.use_graphs and self.dist.capturable()
            net = self.networks["main"]

            def reduced_fb(m, i, e=None):
                """forward / backward of minibatch i with the gradient all-reduce as part of the same call sequence
                (RCCL collectives are graph nodes): one bucket, or — overlap_allreduce — the FC + head gradients
                reduced on RCCL's stream underneath the convolution backward, the convolution gradients behind it."""
                if self.overlap_allreduce and net.late_gradient_offset() > 0:
                    grads, off = net.params.grads, net.late_gradient_offset()
                    self._minibatch_fb(m, clip, True, i=i, epoch=e)
                    w1 = self.dist.all_reduce_sum_async(grads[off:])
                    net.backward_rest()
                    w2 = self.dist.all_reduce_sum_async(grads[:off])
                    w1.wait()
                    w2.wait()
                else:
                    self._minibatch_fb(m, clip, i=i, epoch=e)
                    self.dist.all_reduce_sum(net.params.grads)

            if (self.dist is None or in_graph) and self.epoch_graph and not self.ragged:   # (ragged: n differs from phase to phase)
                # every minibatch of the epoch reads its slice of the ONE static permutation buffer pushed above: the
                # whole epoch is one captured graph (one launch from the host instead of nmb, one graph instead of one
                # per minibatch index) — with data parallelism over RCCL the all-reduces are nodes of that graph
                def epoch():
                    e = self._gather_epoch(n)
                    for i in range(nmb):
                        if self.dist is None:
                            self._minibatch_fb(min(B, n - i * B), clip, i=i, epoch=e)
                        else:
                            reduced_fb(min(B, n - i * B), i, e)
                        self._minibatch_finish(scale)
                self._run(("epoch", n, clip, scale, self.dist is not None and self.overlap_allreduce), epoch)
                results.append(self.scalar_acc / nmb)
                continue
            for i in range(nmb):
                m = min(B, n - i * B)
                if self.dist is None:
                    self._run(("mb", m, clip, scale, i), lambda: (self._minibatch_fb(m, clip, i=i),
                                                                  self._minibatch_finish(scale)))
                elif in_graph:
                    self._run(("mb_dp", m, clip, scale, i, self.overlap_allreduce),
                              lambda: (reduced_fb(m, i), self._minibatch_finish(scale)))
                elif not self.overlap_allreduce:
                    # a backend whose collectives cannot be captured (gloo): one blocking all-reduce per minibatch
                    # between two graph segments; the Adam step of minibatch i-1 rides in the same graph as
                    # forward/backward of minibatch i
                    if i == 0:
                        self._run(("mb_fb", m, clip, i), lambda: self._minibatch_fb(m, clip, i=i))
                    else:
                        self._run(("mb_fin_fb", m, clip, scale, i),
                                  lambda: (self._minibatch_finish(scale), self._minibatch_fb(m, clip, i=i)))
                    self.dist.all_reduce_sum(self.networks["main"].params.grads)
                    if i == nmb - 1:
                        self._run(("mb_fin", scale), lambda: self._minibatch_finish(scale))
                else:
                    # the same overlap with eager collectives between three graph segments per minibatch
                    grads, off = net.params.grads, net.late_gradient_offset()
                    self._run(("mb_p1", m, clip, i), lambda: self._minibatch_fb(m, clip, True, i=i))
                    w1 = self.dist.all_reduce_sum_async(grads[off:])
                    self._run(("mb_p2", m), net.backward_rest)
                    w2 = self.dist.all_reduce_sum_async(grads[:off]) if off > 0 else None
                    w1.wait()
                    if w2 is not None:
                        w2.wait()
                    self._run(("mb_fin", scale), lambda: self._minibatch_finish(scale))
            results.append(self.scalar_acc / nmb)
        return results

    def post_training_commands(self):
        self.memory.clean()                                               # :310-312
        self._rec_missing = False

    def train(self):
        """ClippedPPOAgent.train (:314-344)."""
        if not self._should_train():
            return None
        alg = self.ap.algorithm
        net = self.networks["main"]
        results = None
        for pass_index in range(alg.num_consecutive_training_steps):
            net.update_target(1.0)                                        # networks['main'].sync() (:326)
            if pass_index > 0:
                # the columns the acting steps recorded are V(s) and the action probabilities under the weights of the
                # ROLLOUT; from the second pass on the reference's fill_advantages sees the weights the first pass
                # trained (and the old policy sync() just copied from them): recompute
                self._rec_missing = True
            self.fill_advantages()
            n = min(self.memory.num_transitions(), alg.num_consecutive_playing_steps.num_steps)
            order = list(range(n))                                        # dataset[:num_steps] (:330-331)
            random.shuffle(order)                                         # shuffle(dataset) (:332)
            results = self.train_network(order, alg.optimization_epochs)
        self.post_training_commands()
        self.training_iteration += 1
        last = results[-1]
        self.signals = {"Surrogate loss": last[0], "Entropy": last[1], "KL Divergence": last[2],
                        "Value Loss": last[4], "Grads (unclipped)": last[5]}
        return results

    # ------------------------------------------------------------------------------ reporting
    def check_status(self):
        """device-side error bits (a gather outside the rollout, invalid kernel input) become exceptions here."""
        bits = int(self.memory.status.item())
        if bits:
            self.memory.status.zero_()
            raise ValueError("rollout buffer kernel reported an out-of-range row (status bits %d)" % bits)
        for net in self.networks.values():
            net.check_status()

    def episode_statistics(self):
        a = self.ep_acc.cpu().numpy()
        n = max(a[0], 1.0)
        return {"episodes": int(a[0]), "mean_return": a[1] / n, "max_return": a[3], "min_return": a[4],
                "mean_length": a[5] / n}
