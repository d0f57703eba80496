"""DQN / Double DQN on one MI355X — host-side mirror of rl_coach/agents/dqn_agent.py (parameter
classes :33-65, DQNAgent.learn_from_batch :81-113), rl_coach/agents/ddqn_agent.py:43 and the
ValueOptimizationAgent pieces it uses (agents/value_optimization_agent.py: choose_action through
the exploration policy, update_transition_priorities_and_get_weights).

Per vector step: stacked states -> online Q network -> epsilon-greedy (host draws, device argmax /
tie-break) -> env.step -> reward filter -> episode statistics -> replay store.
Per update: replay draw (host RNG, reference order) -> device gather -> target-network and online
forward -> rlx_dqn_targets (TD targets + |TD error|) -> Huber/MSE loss -> backward -> TF1 Adam;
prioritized replay gets its priorities from the device TD errors without a host round trip.
"""
import numpy as np
import torch

from .. import _rlx
from ..architectures.head_parameters import DuelingQHeadParameters, QHeadParameters
from ..architectures.scheme_views import SchemeViews
from ..core_types import DeviceBatch, EnvironmentSteps, RunPhase
from ..exploration_policies.e_greedy import EGreedy, EGreedyParameters
from ..memories.non_episodic.experience_replay import ExperienceReplayParameters
from ..memories.non_episodic.prioritized_experience_replay import PrioritizedExperienceReplay
from ..nn.networks import DQNNet
from ..schedules import LinearSchedule
from .vector_agent import AlgorithmParameters, VectorOffPolicyAgent


class DQNAlgorithmParameters(AlgorithmParameters):       # dqn_agent.py:33-40
    def __init__(self):
        super().__init__()
        self.num_steps_between_copying_online_weights_to_target = EnvironmentSteps(10000)
        self.num_consecutive_playing_steps = EnvironmentSteps(4)
        self.discount = 0.99
        self.supports_parameter_noise = True


class DQNNetworkParameters(SchemeViews):                      # dqn_agent.py:43-53 + NetworkParameters defaults
    def __init__(self):
        self.activation_function = 'relu'
        self.embedder_scheme = 'Medium'
        self.middleware_scheme = 'Medium'
        self.optimizer_type = 'Adam'
        self.batch_size = 32
        self.replace_mse_with_huber_loss = True
        self.create_target_network = True
        self.learning_rate = 0.00025
        self.adam_optimizer_beta1 = 0.9
        self.adam_optimizer_beta2 = 0.99
        self.optimizer_epsilon = 0.0001
        self.scale_down_gradients_by_number_of_workers_for_sync_training = True
        self.heads_parameters = [QHeadParameters()]      # [DuelingQHeadParameters()] for dueling DQN
        self.clip_gradients = None                       # ClipByGlobalNorm threshold (base_parameters.py)


class DQNAgentParameters(object):                        # dqn_agent.py:56-66
    def __init__(self):
        self.algorithm = DQNAlgorithmParameters()
        self.exploration = EGreedyParameters()
        self.memory = ExperienceReplayParameters()
        self.network_wrappers = {"main": DQNNetworkParameters()}
        self.exploration.epsilon_schedule = LinearSchedule(1, 0.1, 1000000)
        self.exploration.evaluation_epsilon = 0.05
        self.seed = 0

    @property
    def path(self):
        return 'coach_amd.agents.dqn_agent:DQNAgent'


class DQNAgent(VectorOffPolicyAgent):
    double_dqn = False
    SIGNAL_NAMES = VectorOffPolicyAgent.SIGNAL_NAMES + ["Q"]            # value_optimization_agent.py:36

    def __init__(self, agent_parameters, environment, device=None, dist=None, use_graphs=None):
        super().__init__(agent_parameters, environment, device, dist, use_graphs)
        ep, net = environment.p, self.ap.network_wrappers["main"]
        self.A = ep.num_actions
        self.batch_size = net.batch_size
        obs_shape = tuple(ep.observation_shape) + (self.stack,) if self.image else tuple(ep.observation_shape)
        self.networks = {"main": DQNNet(
            self.device, obs_shape, self.A, activation=net.activation_function,
            embedder=net.embedder_scheme, middleware=net.middleware_scheme,
            learning_rate=net.learning_rate, adam_beta1=net.adam_optimizer_beta1,
            adam_beta2=net.adam_optimizer_beta2, optimizer_epsilon=net.optimizer_epsilon,
            replace_mse_with_huber_loss=net.replace_mse_with_huber_loss, seed=self.ap.seed or 0,
            dueling=isinstance(net.heads_parameters[0], DuelingQHeadParameters),
            head_activation=net.heads_parameters[0].activation_function,
            head_gradient_rescale=net.heads_parameters[0].rescale_gradient_from_head_by_factor,
            clip_gradients=net.clip_gradients)}
        self.memory = self._make_memory(action_dim=None)
        self.exploration_policy = EGreedy(self.A, self.n_env, self.device, self.ap.exploration)
        self.actions = torch.zeros(self.n_env, dtype=torch.int32, device=self.device)
        self.td_errors = torch.zeros(self.batch_size, dtype=torch.float64, device=self.device)
        self.loss_acc = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._finish_init()

    # --------------------------------------------------------------------------------- acting
    def random_actions(self):
        """spaces.DiscreteActionSpace.sample (spaces.py:406-407): np.random.choice(actions) per env."""
        a = np.array([np.random.choice(self.A) for _ in range(self.n_env)], dtype=np.int32)
        self.actions.copy_(self._to_device("rand_act", a, torch.int32))
        return self.actions

    def choose_action(self, states):
        net = self.networks["main"]
        self.exploration_policy.phase = self.phase
        draws = self.exploration_policy.draw()                       # host RNG, per env, in order
        if net.can_act_fused(self.n_env):
            # small MLP, a few envs: Q(s) and the epsilon-greedy choice are ONE launch
            eps, d = self.exploration_policy.stage(draws)
            net.q_act(states, self.n_env, d["u"], d["ra"], d["tie"], eps, self._q_buf(), self.actions)
            return self.actions
        self._run(("q", self.n_env), lambda: self._q_forward(states))
        self.exploration_policy.get_action(self._q_act, draws, self.actions)
        return self.actions

    def _q_buf(self):
        if getattr(self, "_q_act", None) is None or self._q_act.shape != (self.n_env, self.A):
            self._q_act = torch.zeros(self.n_env, self.A, dtype=torch.float32, device=self.device)
        return self._q_act

    def _q_forward(self, states):
        q = self.networks["main"].q_values(states, self.n_env, tag="act")
        self._q_act = q.data.view(self.n_env, self.A)

    # ------------------------------------------------------------------------------- training
    PER_UPDATE_RIDES = True

    def _per_ride_ok(self):
        """the image network's backward pass ends in a deferred-reduction launch (the fused small-MLP update and the
        dueling head's path do not pass through Context.flush_deferred in a way that is guaranteed to run)."""
        net = self.networks["main"]
        return self.image and net._fused is None

    def _learn_device(self, b, weights, per_ride=None):
        net = self.networks["main"]
        net.ctx.per_tail = per_ride
        net.learn_from_batch(b._states["observation"], b._next_states["observation"], self.batch_size,
                             b.actions(), b.rewards(), b.game_overs(), self.ap.algorithm.discount,
                             importance_weights=weights, td_errors=self.td_errors,
                             double_dqn=self.double_dqn, grad_scale=self._grad_scale(),
                             sync=self if self.dist is not None else None, states_pair=b._info.get("states_pair"))
        if net.ctx.per_tail is not None:          # no deferred-reduction launch took it along
            net.ctx.per_tail = None
            _rlx.lib().per_update(*per_ride, _rlx.current_stream())

    def _grad_scale(self):
        netp = self.ap.network_wrappers["main"]
        return self.dist.grad_scale(netp.scale_down_gradients_by_number_of_workers_for_sync_training) \
            if self.dist else 1.0

    def learn_from_batch(self, batch):
        """DQNAgent.learn_from_batch (dqn_agent.py:81-113)."""
        per = isinstance(self.memory, PrioritizedExperienceReplay)
        weights = batch.info("weight") if per else None           # fp64, as rlx_per_sample wrote them
        # update_transition_priorities_and_get_weights: the priority update rides on the backward pass's deferred-reduction
        # launch (rlx_splitk_reduce_jobs_per_update) where the network's backward pass has one; else a launch of its own
        ride = self.memory.priority_update_args(batch.info("idx"), self.td_errors) \
            if per and self.PER_UPDATE_RIDES and self._per_ride_ok() else None
        self._run(("learn", per, ride is not None), lambda: self._learn_device(batch, weights, ride))
        if per and ride is None:
            self.memory.update_priorities(batch.info("idx"), self.td_errors)
        loss = self.networks["main"].loss
        self.signals = {"Loss": loss, "Grads (unclipped)": self.networks["main"].norm}
        return loss


    # ------------------------------------------------------------- one staged record + one graph per env-step
    # A 1-env, batch-32 CartPole step is ~16 launches of 2-35 us kernels: issued one by one the loop is HOST-bound
    # (profiles/r02_c1_host_gpu_breakdown.txt: act() 113 us + train() 45 us of Python / launch overhead against
    # ~90 us of GPU time).  Everything the host decides in a step — the epsilon-greedy draws, where the transition
    # is stored, which transitions are sampled — is DATA: it is packed into one record, shipped with ONE async
    # copy, and the whole step (Q forward, action selection, env step, reward filter, episode statistics, store,
    # gather, update[s]) replays as ONE hipGraph that reads the record.  Call order, host RNG consumption and
    # every kernel are those of act() + train(): the results are bit-identical (tests/test_dqn_agent.py).
    def _step_graph_ok(self):
        from ..memories.non_episodic.experience_replay import ExperienceReplay
        return (self.use_graphs and self.dist is None and self.phase == RunPhase.TRAIN and not self.image
                and type(self.memory) is ExperienceReplay and self.signal_stats is None
                and self.debug_draws is None and self.debug_losses is None
                and hasattr(self.env, "launch_step") and hasattr(self.env, "host_tick"))

    def _step_record(self, k):
        """device record of one step with k updates and its typed views (layout fixed per k)."""
        rec = self._records.get(k) if hasattr(self, "_records") else None
        if rec is None:
            if not hasattr(self, "_records"):
                self._records = {}
            from ..staging import Stager
            n, A, B = self.n_env, self.A, self.batch_size
            off, lay = 0, {}
            for name, count, dt, size in (("u", n, torch.float64, 8), ("tie", n * A, torch.float64, 8),
                                          ("ra", n, torch.int32, 4), ("dst", n, torch.int32, 4),
                                          ("rows", max(k, 1) * B, torch.int32, 4)):
                off = (off + 15) // 16 * 16
                lay[name] = (off, count, dt, size)
                off += count * size
            st = Stager(((off + 15) // 16 * 16,), torch.uint8, self.device, depth=32)
            views = {name: st.dst[o:o + c * sz].view(dt) for name, (o, c, dt, sz) in lay.items()}
            views["tie"] = views["tie"].view(n, A)
            views["rows"] = views["rows"].view(max(k, 1), B)
            rec = self._records[k] = dict(stager=st, lay=lay, views=views, host=np.zeros(st.dst.numel(), np.uint8))
        return rec

    def _observe_device(self, dst_rows):
        """reward filter, episode totals, the transition store (rows dst_rows) and the envs' next current state: one
        launch for a few vector-observation envs (rlx_observe_step), else the four launches it stands for."""
        mem, env, alg, s = self.memory, self.env, self.ap.algorithm, _rlx.current_stream()
        has_clip = alg.reward_clipping is not None
        lo, hi = alg.reward_clipping if has_clip else (0.0, 0.0)
        if self.n_env * mem.obs_dim * 4 <= (1 << 16):
            import ctypes
            d = _rlx.ObserveDesc()
            p = lambda t: t.data_ptr()
            d.reward, d.filtered_reward = p(env.reward), p(self.filtered_reward)
            d.reward_rescale, d.has_clip, d.clip_low, d.clip_high = float(alg.reward_rescale), int(has_clip), lo, hi
            d.game_over = d.stored_game_over = p(env.game_over)
            d.ep_return, d.ep_len, d.acc = p(self.ep_return), p(self.ep_len), p(self.ep_acc)
            d.last_return, d.last_len = p(self.last_return), None
            d.actions, d.action_row_bytes = p(self.actions), self.actions.element_size() * (self.actions[0].numel())
            d.cur_state, d.next_obs, d.reset_obs = p(mem.cur_state), p(env.next_obs), p(env.reset_obs)
            d.obs_row_bytes = mem.obs_dim * 4
            d.mem_action, d.mem_reward, d.mem_game_over = p(mem.action), p(mem.reward), p(mem.game_over)
            d.mem_obs, d.mem_next_obs = p(mem.obs), p(mem.next_obs)
            d.dst_rows, d.mem_rows, d.status, d.n_env = p(dst_rows), mem.rows, p(mem.status), self.n_env
            self.lib.observe_step(ctypes.byref(d), s)
            return
        self.lib.reward_filter(env.reward, self.filtered_reward, self.n_env, alg.reward_rescale, int(has_clip),
                               lo, hi, s)
        self.lib.episode_stats_step(self.filtered_reward, env.game_over, self.ep_return, self.ep_len, self.n_env,
                                    self.ep_acc, self.last_return, None, s)
        mem.store_device(self.actions, self.filtered_reward, env.game_over, env.next_obs, env.reset_obs, dst_rows)

    def _step_body(self, k, start, with_act):
        """the device work of one step: [act, env, store] then updates start .. start+k-1 of the record."""
        v = self._step_record(self._rec_k)["views"]
        mem, s, alg = self.memory, _rlx.current_stream(), self.ap.algorithm
        if with_act:
            net = self.networks["main"]
            if net.can_act_fused(self.n_env):
                net.q_act(mem.current_states(), self.n_env, v["u"], v["ra"], v["tie"], 0.0, self._q_buf(), self.actions)
            else:
                self._q_forward(mem.current_states())
                self.lib.egreedy(self._q_act, self.A, v["u"], v["ra"], v["tie"], 0.0, self.n_env, self.A,
                                 self.actions, s)
            self.env.launch_step()
            self._observe_device(v["dst"])
        B = self.batch_size
        b = mem._batch_buffers(B)
        for j in range(start, start + k):
            mem.gather_device(v["rows"][j], B, b)
            batch = DeviceBatch(B, {"observation": b["state"]}, {"observation": b["next_state"]}, b["action"],
                                b["reward"], b["game_over"], info={"states_pair": b["states_pair"]})
            self._learn_device(batch, None)

    def step_and_train(self):
        """One TRAIN-phase vector step: act() then train(), as one record + one graph replay when the
        configuration allows (uniform vector replay, one GPU), else exactly those two calls."""
        if not self._step_graph_ok():
            self.act()
            return self.train()
        mem, pol, n, B, alg = self.memory, self.exploration_policy, self.n_env, self.batch_size, self.ap.algorithm
        # ---- host half of act(): same order of host RNG draws and bookkeeping
        mem.commit_pending()
        pol.phase = self.phase
        eps, u, ra, tie = pol.draw()
        self.env.host_tick()
        dones_host, ended, any_ended, all_ended = self._episode_ends_host()
        row0 = mem.reserve_step(defer=not all_ended)
        self._after_step_host(dones_host, ended, any_ended, all_ended, True)
        # ---- host half of train(): the draws of every update of every phase, target-copy positions
        phases = self._training_phases_due()
        draws, target_after = [], []
        for _ in range(phases):
            phase_draws = [mem.physical_rows(mem.draw(B)) for _ in range(self._training_steps_this_phase())]
            for d in phase_draws:
                self.training_iteration += 1
                draws.append(d)
                if self.networks["main"].target is not None and self._should_update_online_weights_to_target():
                    target_after.append(len(draws))            # copy online -> target after this many updates
        k = len(draws)
        rec = self._step_record(k)
        self._rec_k = k
        host, lay = rec["host"], rec["lay"]

        def put(name, arr, dt):
            o, c, _, sz = lay[name]
            host[o:o + c * sz] = np.ascontiguousarray(arr, dtype=dt).reshape(-1).view(np.uint8)
        put("u", np.where(u < eps, -1.0, 2.0), np.float64)      # explore iff u < epsilon: decided here, epsilon = 0 there
        put("tie", tie, np.float64)
        put("ra", ra, np.int32)
        put("dst", (row0 + np.arange(n)) % mem.rows, np.int32)
        if k:
            put("rows", np.stack(draws), np.int32)
        rec["stager"].push(host)
        # ---- device: one replay (split only where a target copy falls between two updates of the step)
        cuts = [c for c in target_after if c < k]
        start = 0
        for i, end in enumerate(cuts + [k]):
            with_act = i == 0
            key = ("step", k, start, end - start, with_act)
            self._run(key, lambda s_=start, e_=end, w_=with_act: self._step_body(e_ - s_, s_, w_))
            if end in target_after:
                self.update_target_networks(alg.rate_for_copying_weights_to_target)
            start = end
        loss = self.networks["main"].loss
        self.signals = {"Loss": loss, "Grads (unclipped)": self.networks["main"].norm}
        return loss if k else None


def __getattr__(name):
    # Double DQN lives in ddqn_agent.py, as in the reference; older imports of it from this module keep working
    if name in ("DDQNAgent", "DDQNAgentParameters"):
        from . import ddqn_agent
        return getattr(ddqn_agent, name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
