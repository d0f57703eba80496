"""CPU timing of the REFERENCE's own Clipped-PPO hot path on the BASELINE C2 shapes (build container only:
needs /root/reference).

The REAL `rl_coach.agents.clipped_ppo_agent.ClippedPPOAgent` — built by its own __init__ (EpisodicExperienceReplay,
Categorical exploration, episode buffers, `_should_train` with act_for_full_episodes) and stepped through
LevelManager.step's cycle (observe -> act -> env.step; `train -> fill_advantages -> train_network`,
clipped_ppo_agent.py:157-344) — with the reference's own ObservationStackingFilter / RewardClippingFilter on
84x84 uint8 frames, 6 actions, episodes of 32 steps, minibatches of 64.  TensorFlow cannot be installed, so the
network behind `create_networks` is the repo's numpy oracle of the same topology (two conv towers, VHead, PPOHead;
tests/golden/_oracle_backend.py) — the stand-in tests/golden/make_golden.py::gen_ppo_loop pins to the reference.

One thread, as the reference pins it (OMP_NUM_THREADS=1, coach.py:666; TF intra/inter-op = 1,
graph_manager.py:219-220).  A full C2 iteration (2048 env-steps + 320 updates) takes tens of minutes on one
core, so a BOUNDED sample is timed and scaled: `--play` env-steps of rollout and ONE training phase over them
(1 epoch: play/64 minibatch updates + the chunked value pass of fill_advantages); the per-step and per-update
costs are then scaled to 2048 steps + 32 value chunks + 320 updates.  `--procs N` runs N independent copies
on N cores and reports the aggregate (the reference scales by running N rollout workers).

    python tools/time_reference_cpu_c2.py --procs 8 > profiles/r02_cpu_reference_c2.json
"""
import os

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")
import argparse
import json
import multiprocessing as mp
import platform
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_worker(play, seed, q=None, reps=1):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import _refstub
    _refstub.install()
    import _oracle_backend as OB
    from coach_amd.nn import graph as G, networks as NW
    from oracle.agents import ClippedPPOOracle
    from oracle.synth_env import SynthVecEnv
    from rl_coach.agents.clipped_ppo_agent import ClippedPPOAgent, ClippedPPOAgentParameters
    from rl_coach.base_parameters import TaskParameters
    from rl_coach.core_types import EnvResponse, EnvironmentSteps, RunPhase
    from rl_coach.filters.filter import InputFilter, NoInputFilter, NoOutputFilter
    from rl_coach.filters.observation.observation_stacking_filter import ObservationStackingFilter
    from rl_coach.filters.reward.reward_clipping_filter import RewardClippingFilter
    from rl_coach.spaces import (DiscreteActionSpace, PlanarMapsObservationSpace, RewardSpace, SpacesDefinition,
                                 StateSpace)
    H = W = 84
    A, L, B = 6, 32, 64
    # the C2 network, initialised like the device net (same builder, host only)
    params = G.FlatParams()
    torso, feat = NW.build_torso(params, "main", (H, W, 4), "tanh", 2)
    vh = G.Dense(params, "main/v_head/dense", feat, 1, None, 1, init=G.normalized_columns(1.0))
    ph = G.Dense(params, "main/ppo_head/policy_fc", feat, A, None, 1)
    params.finalize("cpu")
    rng = np.random.RandomState(0)
    for m in (torso, vh, ph):
        m.initialize(rng)
    arrays = params.named_arrays()

    ap = ClippedPPOAgentParameters()
    ap.task_parameters = TaskParameters()
    ap.name = "agent"
    ap.visualization.dump_csv = False
    ap.is_a_highest_level_agent = False
    flt = InputFilter(is_a_reference_filter=False)
    flt.add_observation_filter('observation', 'stacking', ObservationStackingFilter(4))      # Atari chain tail
    flt.add_reward_filter('clipping', RewardClippingFilter(-1.0, 1.0))
    ap.input_filter, ap.output_filter, ap.pre_network_filter = flt, NoOutputFilter(), NoInputFilter()
    ap.network_wrappers['main'].batch_size = B
    ap.algorithm.num_consecutive_playing_steps = EnvironmentSteps(play)
    ap.algorithm.optimization_epochs = 1

    class Agent(ClippedPPOAgent):
        def create_networks(self):
            net = ClippedPPOOracle(arrays, (H, W, 4), A, clip_eps=self.ap.algorithm.clip_likelihood_ratio_using_epsilon,
                                   beta_entropy=self.ap.algorithm.beta_entropy)
            w = OB.PPOWrapper(net)
            w.has_target = True
            return {'main': w}
    agent = Agent(ap)
    agent.set_environment_parameters(SpacesDefinition(
        state=StateSpace({'observation': PlanarMapsObservationSpace(np.array([H, W]), 0, 255)}), goal=None,
        action=DiscreteActionSpace(A), reward=RewardSpace(1)))
    agent.update_log = lambda: None
    env = SynthVecEnv(0, 1, H * W, L, 1234 + seed)
    random.seed(seed)
    np.random.seed(seed)
    frame = lambda a: a.reshape(H, W).copy()
    resp = EnvResponse(next_state={'observation': frame(env.reset()[0])}, reward=0, game_over=False)
    agent.reset_internal_state()
    agent.phase = RunPhase.TRAIN
    reset_required, first = False, None
    t_act = t_train = 0.0
    steps = updates = 0
    done_reps = []
    # fill_advantages (one value pass over the whole rollout, once per training phase) is timed apart from the minibatch
    # loop: a full C2 iteration has ONE of them per 320 minibatch updates, the sample one per play/64
    fill = {"t": 0.0}
    inner_fill = agent.fill_advantages

    def timed_fill(batch):
        f0 = time.perf_counter()
        out = inner_fill(batch)
        fill["t"] += time.perf_counter() - f0
        return out
    agent.fill_advantages = timed_fill
    while len(done_reps) < reps:
        t0 = time.perf_counter()
        if reset_required:
            agent.reset_internal_state()
            resp = EnvResponse(next_state={'observation': frame(first)}, reward=0, game_over=False)
            reset_required = False
        agent.observe(resp)
        agent.act()
        nxt, rst, rew, done = env.step()
        resp = EnvResponse(next_state={'observation': frame(nxt[0])}, reward=float(rew[0]), game_over=bool(done[0]))
        if resp.game_over:
            agent.observe(resp)
            agent.handle_episode_ended()
            reset_required, first = True, rst[0]
        t1 = time.perf_counter()
        before = agent.training_iteration
        agent.train()
        t2 = time.perf_counter()
        steps += 1
        t_act += t1 - t0
        if agent.training_iteration != before:
            # Agent.training_iteration counts training PHASES; a phase is 1 epoch of steps/64 minibatch updates here
            updates = steps // B
            t_train = t2 - t1
            done_reps.append(dict(steps=steps, updates=updates, t_act=t_act, t_train=t_train, t_fill=fill["t"]))
            t_act, steps, fill["t"] = 0.0, 0, 0.0
    if q is not None:
        q.put(done_reps)
    return done_reps


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--play", type=int, default=128, help="env-steps of the timed sample (a multiple of 64 and of 32)")
    p.add_argument("--procs", type=int, default=1)
    p.add_argument("--reps", type=int, default=1, help="timed samples per process; the MEDIAN is reported "
                                                        "(BASELINE.md section 3: median of 5)")
    a = p.parse_args()
    t0 = time.perf_counter()
    if a.procs == 1:
        results = [one_worker(a.play, 0, None, a.reps)]
    else:
        q = mp.Queue()
        ps = [mp.Process(target=one_worker, args=(a.play, i, q, a.reps)) for i in range(a.procs)]
        for x in ps:
            x.start()
        results = [q.get() for _ in ps]
        for x in ps:
            x.join()
    wall = time.perf_counter() - t0
    per_proc, all_reps = [], []
    for reps in results:
        cand = []
        for r in reps:
            t_step = r["t_act"] / r["steps"]
            # one training phase over `steps` transitions = fill_advantages over them + steps/64 minibatch updates
            # (1 epoch); a full iteration = 2048 steps + fill_advantages over 2048 transitions + 320 minibatch updates
            t_update = (r["t_train"] - r["t_fill"]) / r["updates"]
            t_fill = r["t_fill"] * 2048.0 / r["steps"]
            t_iter = 2048 * t_step + t_fill + 320 * t_update
            cand.append(dict(env_steps_per_s=2048 / t_iter, updates_per_s=320 / t_iter, s_per_env_step=t_step,
                             s_per_update=t_update, s_fill_advantages_2048=t_fill))
        cand.sort(key=lambda c: c["env_steps_per_s"])
        per_proc.append(cand[len(cand) // 2])                  # the median repetition of this process
        all_reps.append([round(c["env_steps_per_s"], 3) for c in cand])
    results = [r[0] for r in results]
    agg = sum(x["env_steps_per_s"] for x in per_proc)
    print(json.dumps({
        "metric": "env-steps/sec (+ grad-updates/sec), C2, reference code on CPU",
        "value": round(per_proc[0]["env_steps_per_s"] if a.procs == 1 else agg, 3), "unit": "env-steps/s",
        "grad_updates_per_s": round(sum(x["updates_per_s"] for x in per_proc), 4),
        "kind": "reference", "cores": a.procs,
        "per_process": {k: round(float(np.mean([x[k] for x in per_proc])), 5) for k in per_proc[0]},
        "sample": "per process: %d env-steps of rollout (Agent.observe / act with the reference's stacking + reward "
                  "clipping filters, one policy forward per step) and one training phase over them (fill_advantages, timed "
                  "apart, + train_network, 1 epoch = %d minibatch updates of 64 incl. the per-minibatch old-policy pass), "
                  "scaled to a full C2 iteration = 2048 env-steps + one fill_advantages over 2048 transitions + 320 "
                  "updates; numpy oracle (1 thread) as the "
                  "network backend; %d repetitions, median reported; wall %.0f s"
                  % (results[0]["steps"], results[0]["updates"], a.reps, wall),
        "repetitions_per_process": a.reps, "env_steps_per_s_of_every_repetition": all_reps,
        "protocol": "BASELINE.md section 3: >= 10^3 env-steps per repetition for image configs, median of the "
                    "repetitions, time.perf_counter, heat-up excluded (PPO has none)",
        "host": platform.processor() or platform.machine(), "host_cores": os.cpu_count(),
        "where": "build container (NOT the GPU box: /root/reference does not exist there)"}))


if __name__ == "__main__":
    main()
