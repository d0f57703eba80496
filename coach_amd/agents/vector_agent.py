"""Shared machinery of the off-policy vector agents: N lockstep envs per GPU, device replay,
training cadence, target-network cadence, episode statistics, hipGraph replay of the update.

Mirrors the scheduling half of rl_coach/agents/agent.py for N envs stepping in lockstep:
  * Agent.act (:775-840): HEATUP draws uniform random actions (:838-840), TRAIN asks the agent;
  * Agent.observe / observe_transition (:905-973): reward filters, episode accumulation, store;
  * Agent._should_train / train (:662-770): a training phase every
    ``num_consecutive_playing_steps`` env-steps, ``num_consecutive_training_steps`` sampled batches
    per phase; target-network cadence by EnvironmentSteps or TrainingSteps (:640-660).
One vector step advances the step counters by n_env, so a vector step opens
n_env / num_consecutive_playing_steps training phases — the updates-per-env-step ratio of the
reference is preserved; with n_env = 1 the call order, including every host RNG
draw, is the reference's.
"""
import os
import random

import numpy as np
import torch

from .. import _rlx
from ..core_types import EnvironmentSteps, RunPhase, TrainingSteps


class no_gc_while_capturing(object):
    """Python's cyclic garbage collector must not run inside a stream capture: a dead cycle that owns device memory or
    another hipGraph (an agent of an earlier run, with its closures) would be freed in the middle of the capture — a
    device free / graph destroy there aborts the process.  Reference-counted frees of the capture's own temporaries are
    unaffected; whatever garbage accumulates is collected after the capture."""

    def __enter__(self):
        import gc
        self._was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self._was:
            gc.enable()
        return False


def capture(fn):
    from ..distributed import quiesce_before_capture
    quiesce_before_capture()         # (eager RCCL collectives still held by the process group's watchdog: see there)
    g = torch.cuda.CUDAGraph()
    # thread_local: with data parallelism RCCL's watchdog thread polls events while we capture; only
    # calls made by THIS thread may invalidate the capture
    with no_gc_while_capturing(), torch.cuda.graph(g, capture_error_mode="thread_local"):
        fn()
    return g


class Branches(object):
    """Fork / join of side streams for INDEPENDENT parts of one update (SAC: the policy, V and Q passes touch
    different networks until their losses meet).  Works the same eagerly and inside a stream capture, where the event
    edges become the dependencies of a hipGraph with parallel branches.  Python still issues the calls one after
    the other; only the device may overlap them."""

    def __init__(self, n):
        self.streams = [torch.cuda.Stream() for _ in range(n)]

    def fork(self):
        ev = torch.cuda.Event()
        ev.record()
        for s in self.streams:
            s.wait_event(ev)

    def on(self, i):
        return torch.cuda.stream(self.streams[i])

    @staticmethod
    def mark():
        """an event behind everything issued so far on the CURRENT stream"""
        ev = torch.cuda.Event()
        ev.record()
        return ev

    @staticmethod
    def after(ev):
        torch.cuda.current_stream().wait_event(ev)

    def join(self):
        for s in self.streams:
            ev = torch.cuda.Event()
            ev.record(s)
            torch.cuda.current_stream().wait_event(ev)


class _SegmentedCapture(object):
    """A call sequence with collectives in it: every stretch between two all-reduces becomes its own
    hipGraph; the collectives stay eager calls between the replays (RCCL is not captured)."""

    def __init__(self):
        self.items, self._g, self._ctx = [], None, None

    def begin(self):
        self._g = torch.cuda.CUDAGraph()
        self._nogc = no_gc_while_capturing()
        self._nogc.__enter__()
        self._ctx = torch.cuda.graph(self._g, capture_error_mode="thread_local")
        self._ctx.__enter__()

    def end(self, *exc):
        try:
            self._ctx.__exit__(*(exc or (None, None, None)))
        finally:
            self._nogc.__exit__(None, None, None)
        self.items.append(("graph", self._g))

    def split(self, flat):
        self.end()
        self.items.append(("all_reduce", flat))
        self.begin()


class GraphRunner(object):
    """Run a pure-device call sequence eagerly once (allocations), capture it into a hipGraph the
    second time, replay afterwards.  use_graphs=False (constructor argument, or BasicRLGraphManager.use_graphs) keeps everything eager.  With data parallelism
    (`self.dist`) the sequence is cut at every `_allreduce` call into graph segments."""

    _segcap = None

    def _collectives_in_graph(self):
        """data parallel over RCCL: the all-reduce is a node of the captured update (GradientSync.capturable)."""
        d = getattr(self, "dist", None)
        return d is not None and d.capturable()

    def _allreduce(self, flat):
        """Sum `flat` over the ranks: an eager (or captured — RCCL) collective, or a segment boundary while capturing
        with a backend that cannot be captured."""
        if self._segcap is not None:
            self._segcap.split(flat)
        elif getattr(self, "dist", None) is not None:
            self.dist.all_reduce_sum(flat)

    all_reduce_sum = _allreduce          # the `sync` object handed to network-level updates

    def _run_segmented(self, key, fn):
        segs = self._graphs.get(key)
        if segs is None:
            if key not in self._warm:
                self._warm.add(key)
                return fn()
            from ..distributed import quiesce_before_capture
            quiesce_before_capture()         # (once per captured sequence: no collective runs between its segments' captures)
            cap = self._segcap = _SegmentedCapture()
            cap.begin()
            try:
                fn()
            except BaseException:
                import sys
                self._segcap = None
                cap.end(*sys.exc_info())
                raise
            self._segcap = None
            cap.end()
            segs = self._graphs[key] = cap.items
        for kind, obj in segs:
            if kind == "graph":
                obj.replay()
            else:
                self.dist.all_reduce_sum(obj)

    def _init_graphs(self, use_graphs=None):
        self.use_graphs = True if use_graphs is None else bool(use_graphs)
        self._graphs, self._warm = {}, set()

    _inline_run = False      # set while a captured sequence runs sub-sequences that would otherwise be graphs of their own

    def _run(self, key, fn):
        if not self.use_graphs or self._inline_run:
            return fn()
        if getattr(self, "dist", None) is not None and not self._collectives_in_graph():
            return self._run_segmented(key, fn)
        g = self._graphs.get(key)
        if g is not None:
            return g.replay()
        if key not in self._warm:
            self._warm.add(key)
            return fn()
        torch.cuda.synchronize()
        self._graphs[key] = capture(fn)
        return self._graphs[key].replay()


class AlgorithmParameters(object):                       # base_parameters.py:170-230 (hot-path fields)
    def __init__(self):
        self.discount = 0.99
        self.num_consecutive_playing_steps = EnvironmentSteps(1)
        self.num_consecutive_training_steps = 1
        self.heatup_using_network_decisions = False
        self.num_steps_between_copying_online_weights_to_target = TrainingSteps(0)
        self.rate_for_copying_weights_to_target = 1.0
        self.act_for_full_episodes = False
        self.reward_clipping = None                      # RewardClippingFilter bounds or None
        self.reward_rescale = 1.0                        # RewardRescaleFilter factor


class _HostDrawsAhead(object):
    """The host-RNG side of one Agent.train call — per phase the index draws of every batch (agent.py:726), then per update its
    own draws (TD3's smoothing noise, SAC's three normal arrays: `_draw_update_host`) — made IN ORDER by a producer thread a
    bounded number of updates ahead of the loop that ships them.  Nothing else consumes the host streams inside train() and
    the uniform replay does not change there, so the values and their order are those of the loop drawing them itself
    (tests/test_host_draws_ahead.py: weights and generator state bit-identical); numpy's legacy generators fill arrays
    without the GIL, so SAC's 13 056 normals per update (119 us on one core — as long as the device's whole update) come
    off the launching thread."""

    def __init__(self, items, n, depth=32):
        """items: an iterator over the n per-update host items, in the order the consuming loop would make them"""
        import queue
        import threading
        self._q, self._stop = queue.Queue(maxsize=depth), False
        self._t = threading.Thread(target=self._fill, args=(items, n), daemon=True)
        self._t.start()

    def __next__(self):
        return self.get()

    def _fill(self, items, n):
        import queue
        try:
            for _ in range(n):
                item = next(items)
                while not self._stop:
                    try:
                        self._q.put(item, timeout=0.05)
                        break
                    except queue.Full:
                        pass
                if self._stop:
                    return
        except BaseException as e:          # surfaces in the consuming thread
            self._q.put(e)

    def get(self):
        item = self._q.get()
        if isinstance(item, BaseException):
            raise item
        return item

    def close(self):
        self._stop = True
        self._t.join()


class VectorOffPolicyAgent(GraphRunner):
    continuous = False
    # signals every agent registers (agent.py:187-192), in registration order; subclasses append their own.
    # 'Reward' / 'Shaped Reward' are per-step signals and have no episode columns.
    SIGNAL_NAMES = ["Loss", "Learning Rate", "Grads (unclipped)", "Discounted Return"]
    signal_stats = None              # DeviceSignals once enable_signal_statistics() is called (CSV logging)

    def __init__(self, agent_parameters, environment, device=None, dist=None, use_graphs=None):
        self.ap = agent_parameters
        self.env = environment
        self.device = device or environment.device
        self.dist = dist if (dist is not None and dist.enabled) else None
        self.lib = _rlx.lib()
        self._init_graphs(use_graphs)
        ep = environment.p
        self.n_env = ep.num_envs
        self.image = ep.kind == "image"
        self.stack = 4 if self.image else None
        self.L = ep.episode_length
        if self.ap.seed is not None:                     # agents/agent.py:49-55
            random.seed(self.ap.seed)
            np.random.seed(self.ap.seed)
        self.phase = RunPhase.HEATUP
        self.total_steps_counter = 0
        self.training_iteration = 0
        self.last_training_phase_step = 0
        self.last_target_network_update_step = 0
        self.current_episode_steps_counter = 0                  # steps of env 0's running episode
        self._episode_steps = np.zeros(self.n_env, dtype=np.int64)     # ... of every env's
        self._unconsumed_episode_lengths = []                   # finished episodes no training phase has used yet
        self._episode_just_ended = False
        dev, n = self.device, self.n_env
        self.filtered_reward = torch.zeros(n, dtype=torch.float32, device=dev)
        self.ep_return = torch.zeros(n, dtype=torch.float64, device=dev)
        self.ep_len = torch.zeros(n, dtype=torch.int32, device=dev)
        self.ep_acc = torch.zeros(8, dtype=torch.float64, device=dev)
        self.lib.episode_stats_init(self.ep_return, self.ep_len, n, self.ep_acc, _rlx.current_stream())
        self.signals = {}
        self.last_return = torch.zeros(n, dtype=torch.float64, device=dev)     # return of each env's last episode
        self._episode_log = []           # (env, length, pinned copy of last_return, event) per finished episode
        from ..staging import StagerCache
        self._stagers = StagerCache(self.device)
        self.debug_draws = None          # tests set these to lists to record every replay draw /
        self.debug_losses = None         # every update's loss (forces a sync per update)

    # ------------------------------------------------------------------ helpers for subclasses
    def _finish_init(self):
        """after the networks exist: rank-offset sampling seed (coach.py:746) and the first obs."""
        if self.dist is not None and self.ap.seed is not None:
            random.seed(self.ap.seed + self.dist.rank)
            np.random.seed(self.ap.seed + self.dist.rank)
        self.memory.reset(self.env.reset_internal_state())

    def _make_memory(self, action_dim=None):
        mp = self.ap.memory
        from ..memories.non_episodic.prioritized_experience_replay import (
            PrioritizedExperienceReplay, PrioritizedExperienceReplayParameters)
        from ..memories.non_episodic.experience_replay import ExperienceReplay
        from ..memories.episodic.episodic_experience_replay import (
            EpisodicExperienceReplay, EpisodicExperienceReplayParameters)
        ep = self.env.p
        kw = dict(device=self.device, n_env=self.n_env, observation_shape=ep.observation_shape,
                  stack=self.stack, action_dim=action_dim,
                  min_episode_length=getattr(ep, "min_episode_length", self.L))
        if isinstance(mp, PrioritizedExperienceReplayParameters):
            return PrioritizedExperienceReplay(mp.max_size, mp.alpha, mp.beta, mp.epsilon,
                                               mp.allow_duplicates_in_batch_sampling,
                                               exact_pow=getattr(mp, "exact_pow", "device"), **kw)
        if isinstance(mp, EpisodicExperienceReplayParameters):
            return EpisodicExperienceReplay(mp.max_size, mp.allow_duplicates_in_batch_sampling,
                                            n_step=getattr(mp, "n_step", -1), discount=self.ap.algorithm.discount,
                                            max_episode_length=self.L, **kw)
        return ExperienceReplay(mp.max_size, mp.allow_duplicates_in_batch_sampling, **kw)

    def _to_device(self, key, array, dtype):
        """host draws -> a static device buffer through a ring of pinned staging slots."""
        return self._stagers.push(key, array, dtype)

    # --------------------------------------------------------------------------------- acting
    def random_actions(self):
        raise NotImplementedError

    def choose_action(self, states):
        raise NotImplementedError

    def act(self):
        """One vector step of LevelManager.step (level_manager.py:215-269) for n_env envs.

        Order of the reference, kept: (1) observe the PREVIOUS response — its transition enters the
        memory now, after the train() that followed the previous step (:236-238); (2) choose an action;
        (3) env.step; (4) a terminal response is observed at once (:260-264), a non-terminal one waits
        for the next call.  The rows of this step are written to the ring immediately either way; only
        their visibility to sampling is deferred (memory.commit_pending)."""
        alg = self.ap.algorithm
        s = _rlx.current_stream()
        record = self.phase != RunPhase.TEST
        if record:
            self.memory.commit_pending()
        states = self.memory.current_states()
        if self.phase == RunPhase.HEATUP and not alg.heatup_using_network_decisions:
            actions = self.random_actions()                                    # agent.py:838-840
        else:
            actions = self.choose_action(states)
        next_obs, reset_obs, reward, game_over = self.env.step(actions)
        has_clip = alg.reward_clipping is not None
        lo, hi = alg.reward_clipping if has_clip else (0.0, 0.0)
        self.lib.reward_filter(reward, self.filtered_reward, self.n_env, alg.reward_rescale,
                               int(has_clip), lo, hi, s)
        if record:
            self.lib.episode_stats_step(self.filtered_reward, game_over, self.ep_return, self.ep_len,
                                        self.n_env, self.ep_acc, self.last_return, None, s)
        dones_host, ended, any_ended, all_ended = self._episode_ends_host()
        stored = self._stored_game_over(game_over)
        # a terminal response is observed at once; the others at the start of the next step.  With several envs
        # the rows of a step become visible together, at once only when EVERY env's episode ended on it.
        self.memory.store(actions, self.filtered_reward, stored, next_obs, reset_obs, record=record,
                          dones=None if stored is game_over else game_over,
                          defer=not all_ended, episode_end=any_ended, dones_host=dones_host)
        return self._after_step_host(dones_host, ended, any_ended, all_ended, record)

    def _episode_ends_host(self):
        """which envs finished on the step just taken: a host fact (the env front end's `dones_host`; lockstep envs
        without one end together after L steps) — no device sync."""
        self.env.total_steps += self.n_env
        self._episode_steps += 1
        dones_host = getattr(self.env, "dones_host", None)
        if dones_host is None:
            dones_host = self._episode_steps >= self.L
        ended = np.nonzero(dones_host)[0]
        return dones_host, ended, ended.size > 0, ended.size == self.n_env

    def _after_step_host(self, dones_host, ended, any_ended, all_ended, record):
        self._episode_just_ended = any_ended
        self.current_episode_steps_counter = int(self._episode_steps[0]) if not dones_host[0] else 0
        if any_ended:
            self.ended_episode_lengths = self._episode_steps[ended].copy()    # one entry per finished env
            self.last_episode_steps = int(self.ended_episode_lengths[-1])
            self._episode_steps[ended] = 0
            if self.phase == RunPhase.TRAIN:       # heat-up episode ends open no training phase later
                self._unconsumed_episode_lengths.extend(int(x) for x in self.ended_episode_lengths)
            if record and self.signal_stats is not None:
                self._note_finished_episodes(ended)
            self.handle_episode_ended()
        if self.phase != RunPhase.TEST:                                        # agent.py:832-834
            self.total_steps_counter += self.n_env
        return all_ended

    def _stored_game_over(self, game_over):
        return game_over

    # ------------------------------------------------------------------- signals / episode log
    def enable_signal_statistics(self):
        """Collect what Agent.update_log writes per episode (agent.py:509-556): the statistics of every registered
        signal and each finished episode's reward and length.  Off by default: it costs one small launch per update
        and one device->host copy per logged episode."""
        if self.signal_stats is None:
            from ..signals import DeviceSignals
            self.signal_stats = DeviceSignals(self.SIGNAL_NAMES, self.device)
        return self.signal_stats

    def _note_finished_episodes(self, ended):
        """asynchronous copy of the finished envs' episode returns; read when the rows are written."""
        host = torch.empty(self.n_env, dtype=torch.float64, pin_memory=True)
        host.copy_(self.last_return, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        alg = self.ap.algorithm
        for e, length in zip(ended.tolist(), self.ended_episode_lengths.tolist()):
            self._episode_log.append((int(e), int(length), host, ev))
            # handle_episode_ended: every transition's n-step discounted return is a 'Discounted Return' sample
            dr = self.memory.episode_discounted_returns(int(e), int(length), alg.discount, getattr(alg, "n_step", -1))
            self.signal_stats.accumulate({"Discounted Return": dr})

    def pop_finished_episodes(self):
        """[(env, length, total shaped reward)] of the episodes finished since the last call."""
        out = []
        for e, length, host, ev in self._episode_log:
            ev.synchronize()
            out.append((e, length, float(host[e])))
        self._episode_log = []
        return out

    def _accumulate_signals(self):
        st = self.signal_stats
        st.accumulate({k: (v if v.dim() else v.view(1)) for k, v in self.signals.items()
                       if isinstance(v, torch.Tensor)})
        lr = getattr(next(iter(self.ap.network_wrappers.values())), "learning_rate", None)
        if lr is not None:
            st.add_host_sample("Learning Rate", lr)                  # curr_learning_rate.add_sample (agent.py:753-754)

    def handle_episode_ended(self):
        pass

    # ------------------------------------------------------------------------------- training
    def _should_update_online_weights_to_target(self):                        # agent.py:640-660
        m = self.ap.algorithm.num_steps_between_copying_online_weights_to_target
        counter = self.training_iteration if isinstance(m, TrainingSteps) else self.total_steps_counter
        if not isinstance(m, (TrainingSteps, EnvironmentSteps)):
            raise ValueError("The num_steps_between_copying_online_weights_to_target parameter should be "
                             "either EnvironmentSteps or TrainingSteps. Instead it is {}".format(type(m)))
        if (counter - self.last_target_network_update_step) >= m.num_steps:
            self.last_target_network_update_step = counter
            return True
        return False

    def _training_phases_due(self):
        """agent.py:662-699 for a step counter that advances n_env at a time."""
        steps = self.ap.algorithm.num_consecutive_playing_steps.num_steps
        # what the reference's memory holds at this point: an episodic memory receives an episode only
        # when it ends (agent.py:576-584), so its num_transitions() counts complete episodes
        mem = self.memory
        stored = mem.num_transitions_in_complete_episodes() if hasattr(mem, "num_transitions_in_complete_episodes") \
            else mem.num_transitions()
        if stored <= 0:
            return 0
        gap = self.total_steps_counter - self.last_training_phase_step
        if gap < steps:
            return 0
        if self.ap.algorithm.act_for_full_episodes:
            # wait_for_full_episode (agent.py:692-693): one phase per episode that finished since the last
            # phase; each env's finished episode is one reference episode end
            if not self._unconsumed_episode_lengths:
                return 0
            self._phase_episode_lengths = self._unconsumed_episode_lengths
            self._unconsumed_episode_lengths = []
            self.last_training_phase_step = self.total_steps_counter
            return len(self._phase_episode_lengths)
        # the reference opens ONE phase per check and resets the marker to the current step
        # (:673-677); a vector step is n_env checks' worth of env-steps
        due = min(gap // steps, max(1, self.n_env // steps))
        self.last_training_phase_step = self.total_steps_counter
        return due

    _staged = None            # device views of the current update's record (see _update_record), or None
    # the host-RNG side of a train() call on a producer thread (_HostDrawsAhead): same values, measured SLOWER — C5 953 - 1087
    # against 727 ms per step, C4 2521 against 2398 (profiles/r06_ab_host_draws_ahead.txt): every hand-over of the GIL between
    # the launching thread and the producer is a futex wake-up, and there are several per update.  Off.
    HOST_DRAWS_AHEAD = False
    # K consecutive updates of a train() call as ONE captured graph over ONE staged record of K rows / K sets of host draws
    # (_train_chunk): one blit and one replay per K updates instead of one record, one gather launch and one or two replays
    # per update — the TD3 / SAC loops are bound by the launching thread, not by the device (C4: 80.9 us of kernels in a
    # 94.8 us update).  0 / 1: every update on its own.
    UPDATE_CHUNK = 8

    def _update_record_fields(self):
        """[(name, shape, dtype)] of the host draws one update consumes (TD3: the smoothing noise, SAC: the three
        normal draws; DDPG: [] — the sampled rows alone are the record), or None: the agent stages nothing (its memory
        ships the rows itself)."""
        return None

    def _draw_update_host(self):
        """{name: array} — the host draws of ONE update, made exactly as learn_from_batch makes them."""
        return {}

    def _update_record(self):
        """A RecordStager [rows | the update's host draws] when the agent has per-update draws and the memory takes
        device rows (uniform / episodic replay; prioritized replay samples on the device), else None."""
        rec = getattr(self, "_update_rec", False)
        if rec is False:
            rec = None
            fields = self._update_record_fields()
            import inspect
            if fields is not None and "rows_dev" in inspect.signature(self.memory.collate).parameters:
                from ..staging import RecordStager
                rec = RecordStager([("rows", (self.batch_size,), torch.int32)] + list(fields), self.device)
            self._update_rec = rec
        return rec

    _mix_rate = None          # set around learn_from_batch when a soft target update follows the update
    _mixed = frozenset()

    def update_target_networks(self, rate):
        for net in self.networks.values():
            if net.target is not None:
                net.update_target(rate)

    def train(self):
        """Agent.train (agent.py:701-770): returns the loss of the last update run (device scalar)
        or None when no training phase was due."""
        if self.phase != RunPhase.TRAIN:
            return None
        phases = self._training_phases_due()
        if phases == 0:
            return None
        alg = self.ap.algorithm
        B = self.batch_size
        losses = []
        rec = self._update_record()
        steps_list = [self._training_steps_this_phase() for _ in range(phases)]

        def host_items():
            """(drawn batch, the update's host draws or None) of every update of this call, in the reference's order on the host
            streams: every batch of a phase is drawn first (agent.py:726), then each update's own draws where
            learn_from_batch would make them; a collated DeviceBatch aliases the memory's static buffers, so collation is
            per batch (below)."""
            for steps in steps_list:
                draws = [self.memory.draw(B) for _ in range(steps)]
                for d in draws:
                    yield d, (self._draw_update_host() if rec is not None else None)
        total = sum(steps_list)
        from ..memories.non_episodic.experience_replay import ExperienceReplay
        # the whole host-RNG side of the call on a producer thread: the uniform / episodic replay's draw() is host-only (the
        # prioritized one launches its descent: excluded), no store happens
        # inside train(), and nothing else touches the host streams until the last update has its draws
        ahead = _HostDrawsAhead(host_items(), total) \
            if rec is not None and self.HOST_DRAWS_AHEAD and total >= 8 and type(self.memory).draw is ExperienceReplay.draw else None
        source = ahead if ahead is not None else host_items()
        K = int(self.UPDATE_CHUNK or 0)
        chunked = 0
        try:
            if K > 1 and rec is not None and self.use_graphs and self.dist is None and self.signal_stats is None and \
                    self.debug_draws is None and self.debug_losses is None:
                while total - chunked >= K:
                    losses.append(self._train_chunk(source, K, B, alg))
                    chunked += K
            for _ in range(total - chunked):
                d, host = next(source)
                self.training_iteration += 1
                if rec is not None:
                    # everything this update needs from the host — the sampled rows AND its random draws — is ONE record, one copy
                    self._staged = rec.push(rows=self.memory.physical_rows(d), **host)
                    batch = self.memory.collate(d, B, rows_dev=self._staged["rows"])
                else:
                    batch = self.memory.collate(d, B)
                if self.debug_draws is not None:      # sampled logical indices / PER leaves
                    self.debug_draws.append(batch.info("idx").cpu().numpy().copy()
                                            if "idx" in batch._info else np.asarray(d).copy())
                # is a target update due after this update (agent.py:640-660)?  Known before it runs: a network whose
                # Adam step is part of the update mixes its target in the same pass (learn_from_batch records which)
                mix = any(n.target is not None for n in self.networks.values()) and \
                    self._should_update_online_weights_to_target()
                self._mix_rate = alg.rate_for_copying_weights_to_target if mix else None
                self._mixed = set()
                loss = self.learn_from_batch(batch)
                if self.signal_stats is not None:
                    self._accumulate_signals()
                if self.debug_losses is not None:
                    self.debug_losses.append(float(loss.sum().item()))
                losses.append(loss)
                if mix:
                    for name, net in self.networks.items():
                        if net.target is not None and name not in self._mixed:
                            net.update_target(self._mix_rate)
                    self._target_updated_since_log = True      # 'Update Target Network' column (agent.py:760)
                self._mix_rate = None
        finally:
            if ahead is not None:
                ahead.close()
        self._staged = None
        # the loss of the last update of the phase(s) (a device scalar; no per-update host sync or add)
        return losses[-1] if losses else None

    def _schedule_phase(self, iteration):
        """what of `training_iteration` decides the SHAPE of the next updates (TD3: the delayed actor step) — part of the key
        of a chunk's graph."""
        return 0

    def _chunk_record(self, K):
        recs = self.__dict__.setdefault("_chunk_recs", {})
        r = recs.get(K)
        if r is None:
            from ..staging import RecordStager
            fields = [("rows", (K, self.batch_size), torch.int32)] + \
                [(n, (K,) + tuple(shape), dt) for n, shape, dt in self._update_record_fields()]
            r = recs[K] = RecordStager(fields, self.device)
        return r

    def _train_chunk(self, source, K, B, alg):
        """The next K updates of this train() call as one graph replay: their sampled rows and host draws (made in the
        loop's order) go out as ONE record, the graph runs gather + learn_from_batch (+ the target mixes that are not part of
        an Adam pass) K times on the record's K slices.  Host bookkeeping — the iteration counter, the target-update rule's
        counters (agent.py:640-660), which update is a delayed-policy step — is done here, outside the captured body, in the
        per-update loop's order; the graph is keyed by everything of it that shapes the launches."""
        crec = self._chunk_record(K)
        it0 = self.training_iteration
        has_target = any(n.target is not None for n in self.networks.values())
        mixes, ds = [], []
        for k in range(K):
            d, host = next(source)
            ds.append(d)
            crec.host_views["rows"][k] = self.memory.physical_rows(d)
            for name, a in host.items():
                crec.host_views[name][k] = a
            self.training_iteration += 1
            mixes.append(bool(has_target and self._should_update_online_weights_to_target()))
        crec.stager.push(crec.host)
        views = crec.views
        rate = alg.rate_for_copying_weights_to_target
        key = ("chunk", K, tuple(mixes), self._schedule_phase(it0))
        cache = self.__dict__.setdefault("_chunk_loss", {})

        def body():
            self._inline_run = True
            try:
                for k in range(K):
                    self.training_iteration = it0 + k + 1
                    self._staged = {n: v[k] for n, v in views.items()}
                    batch = self.memory.collate(ds[k], B, rows_dev=self._staged["rows"])
                    self._mix_rate = rate if mixes[k] else None
                    self._mixed = set()
                    cache[key] = self.learn_from_batch(batch)
                    if mixes[k]:
                        for name, net in self.networks.items():
                            if net.target is not None and name not in self._mixed:
                                net.update_target(self._mix_rate)
                    self._mix_rate = None
            finally:
                self._inline_run = False
        self._run(key, body)
        self.training_iteration = it0 + K
        self._staged = None
        if any(mixes):
            self._target_updated_since_log = True
        return cache[key]

    def _training_steps_this_phase(self):
        return self.ap.algorithm.num_consecutive_training_steps

    def learn_from_batch(self, batch):
        raise NotImplementedError

    def reset_internal_state(self):
        """GraphManager.reset_internal_state(force_environment_reset=True) + Agent.reset_internal_state
        (graph_manager.py:411-424, agent.py:603-629): every env starts a new episode NOW.  The response
        of the last step was never observed, so its transition is never stored; an episodic memory
        loses the whole unfinished episode (it lived in current_episode_buffer); the per-episode
        accumulators restart."""
        mem = self.memory
        if hasattr(mem, "drop_open_episode"):
            mem.drop_open_episode()
        else:
            mem.drop_pending()
        self.current_episode_steps_counter = 0
        self._episode_steps[:] = 0
        self._episode_just_ended = False
        self.lib.episode_stats_init(self.ep_return, self.ep_len, self.n_env, None, _rlx.current_stream())
        pol = getattr(self, "exploration_policy", None)
        if pol is not None and hasattr(pol, "reset"):
            pol.reset()
        return self.env.reset_internal_state()

    def evaluate_episodes(self, episodes_per_env=1):
        """GraphManager.evaluate (graph_manager.py:491-523) for the env vector: every level is reset,
        then whole episodes are played in TEST phase (greedy / evaluation_epsilon acting, nothing
        stored, training counters frozen); training resumes from a fresh reset afterwards (the last
        evaluation episode ended, so reset_required is set).  The memory is not touched: evaluation
        frames go to a scratch stack (memory.begin_evaluation).  Returns the mean undiscounted episode
        reward over envs."""
        prev, prev_env = self.phase, getattr(self.env, "phase", None)
        self.env.phase = RunPhase.TEST           # GraphManager.phase setter: the environment follows (graph_manager.py:333-344)
        first = self.reset_internal_state()
        self.memory.begin_evaluation(first)
        self.phase = RunPhase.TEST
        total = torch.zeros(self.n_env, dtype=torch.float64, device=self.device)
        finished = np.zeros(self.n_env, dtype=np.int64)          # evaluation episodes completed per env
        try:
            while (finished < episodes_per_env).any():
                self.act()
                active = finished < episodes_per_env                      # envs still inside their quota
                if active.all():
                    total += self.env.reward.double()
                else:
                    total += self.env.reward.double() * torch.from_numpy(active.astype(np.float64)).to(self.device)
                dh = getattr(self.env, "dones_host", None)
                finished += (dh if dh is not None else np.full(self.n_env, self._episode_just_ended)).astype(np.int64)
        finally:
            self.phase = prev
            self.env.phase = prev_env if prev_env is not None else prev
            self.current_episode_steps_counter = 0
            self._episode_steps[:] = 0
            self._unconsumed_episode_lengths = []
            self._episode_just_ended = False
            self.memory.end_evaluation(self.env.reset_internal_state())
        return float(total.mean().item()) / episodes_per_env

    # ------------------------------------------------------------------------------ reporting
    def episode_statistics(self):
        a = self.ep_acc.cpu().numpy()
        n = max(a[0], 1.0)
        return {"episodes": int(a[0]), "mean_return": a[1] / n, "max_return": a[3], "min_return": a[4],
                "mean_length": a[5] / n}

    def check_status(self):
        self.memory.check_status()
        for net in self.networks.values():
            net.check_status()
