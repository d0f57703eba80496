"""Checkpoint / resume of a device agent — the naming of rl_coach/checkpoint.py and
graph_manager.py:616-658 (`<id>_Step-<n>.ckpt`, latest name recorded in `.coach_checkpoint`), holding
what a bit-exact resume of the hot path needs: every network's flat weight / target / Adam buffers,
the replay contents and cursors, filter statistics, schedules, step counters and the host RNG states
(`random`, legacy `np.random`) whose draws select replay indices and exploration actions.
"""
import os
import pickle
import random
import re

import numpy as np
import torch

STATE_FILE = ".coach_checkpoint"                          # checkpoint.py CheckpointStateFile
_NAME = re.compile(r"^(\d+)_Step-(\d+)\.ckpt$")


def _tensors_of(obj, names):
    return {n: getattr(obj, n).detach().cpu() for n in names if isinstance(getattr(obj, n, None), torch.Tensor)}


_MEMORY_TENSORS = ["ring", "fpos", "epoff", "t_fpos", "t_epoff", "cur_state", "obs", "next_obs", "action",
                   "reward", "game_over", "sum_tree", "min_tree", "max_tree", "max_priority",
                   "n_step_discounted_rewards", "act_value", "act_probs"]
_MEMORY_SCALARS = ["cursor", "count", "pending", "committed_total", "steps", "_open", "_list_len",
                   "next_leaf_idx_to_write", "_frames_total", "_episode_steps", "_step_frames",
                   "_steps_written", "_gstep", "_ep_start", "_episodes", "_episode_first_step", "_order", "_order_head", "_order_len"]
_AGENT_SCALARS = ["total_steps_counter", "training_iteration", "last_training_phase_step",
                  "last_target_network_update_step", "current_episode_steps_counter", "last_episode_steps",
                  "_episode_just_ended", "_episode_steps", "_unconsumed_episode_lengths",
                  "_draw_pool", "_draw_pos", "_draw_table_from", "_rec_missing"]      # PPO: host draws made for steps that have not run yet


def agent_state(agent):
    st = {"networks": {}, "memory": {}, "agent": {}, "host_rng": (random.getstate(), np.random.get_state())}
    for name, net in agent.networks.items():
        st["networks"][name] = {
            "weights": net.params.weights.cpu(), "target": None if net.target is None else net.target.cpu(),
            "adam_m": net.adam.m.cpu(), "adam_v": net.adam.v.cpu(), "adam_state": net.adam.state.cpu()}
    mem = agent.memory
    st["memory"]["tensors"] = _tensors_of(mem, _MEMORY_TENSORS)
    import copy
    st["memory"]["scalars"] = {k: copy.deepcopy(getattr(mem, k)) for k in _MEMORY_SCALARS if hasattr(mem, k)}
    st["agent"]["scalars"] = {k: copy.deepcopy(getattr(agent, k)) for k in _AGENT_SCALARS if hasattr(agent, k)}
    st["agent"]["phase"] = agent.phase.name                   # RunPhase of the agent (and its policy)
    # uniforms_all: the exploration draws of the running PPO phase (made at its first step)
    st["agent"]["tensors"] = _tensors_of(agent, ["ep_return", "ep_len", "ep_acc", "uniforms_all"])
    st["env"] = {"tensors": _tensors_of(agent.env, ["obs", "episode", "step_in_episode"]),
                 "total_steps": agent.env.total_steps,
                 "host": {k: getattr(agent.env, k).copy() for k in ("t_host", "dones_host") if hasattr(agent.env, k)}}
    pol = getattr(agent, "exploration_policy", None)
    if pol is not None:
        st["exploration"] = pickle.dumps({k: v for k, v in pol.__dict__.items()
                                          if isinstance(v, (int, float, np.ndarray)) or hasattr(v, "current_value")})
    for k in ("beta",):
        if hasattr(mem, k):
            st["memory"][k] = pickle.dumps(getattr(mem, k))
    if getattr(agent, "norm", None) is not None:
        st["filter"] = _tensors_of(agent.norm, ["sum", "sum_squares", "count", "mean", "std"])
    alg = agent.ap.algorithm
    if hasattr(alg, "clipping_decay_schedule"):
        st["clipping_decay_schedule"] = pickle.dumps(alg.clipping_decay_schedule)
    return st


def load_agent_state(agent, st):
    for name, net in agent.networks.items():
        s = st["networks"][name]
        net.params.weights.copy_(s["weights"])
        if net.target is not None:
            net.target.copy_(s["target"])
        net.adam.m.copy_(s["adam_m"]); net.adam.v.copy_(s["adam_v"]); net.adam.state.copy_(s["adam_state"])
    mem = agent.memory
    for k, t in st["memory"]["tensors"].items():
        if k in ("act_value", "act_probs") and not hasattr(mem, k):
            continue                         # written by a recording agent, restored into one that does not record
        getattr(mem, k).copy_(t)
    import copy
    for k, v in st["memory"]["scalars"].items():
        setattr(mem, k, copy.deepcopy(v))
    if "beta" in st["memory"]:
        mem.beta = pickle.loads(st["memory"]["beta"])
    for k, v in st["agent"]["scalars"].items():
        setattr(agent, k, copy.deepcopy(v))
    if hasattr(mem, "act_value") and "act_value" not in st["memory"]["tensors"]:
        # a checkpoint written without the recorded V(s) / probability columns (an agent that did not record, an older
        # writer): the rows of the restored rollout carry none — its training phase runs the dataset passes
        agent._rec_missing = True
    if "phase" in st["agent"]:
        from .core_types import RunPhase
        agent.phase = RunPhase[st["agent"]["phase"]]
        if getattr(agent, "exploration_policy", None) is not None:
            agent.exploration_policy.phase = agent.phase
    for k, t in st["agent"]["tensors"].items():
        getattr(agent, k).copy_(t)
    for k, t in st["env"]["tensors"].items():
        getattr(agent.env, k).copy_(t)
    agent.env.total_steps = st["env"]["total_steps"]
    for k, v in st["env"].get("host", {}).items():
        getattr(agent.env, k)[...] = v
    if "exploration" in st and getattr(agent, "exploration_policy", None) is not None:
        agent.exploration_policy.__dict__.update(pickle.loads(st["exploration"]))
    if "filter" in st and getattr(agent, "norm", None) is not None:
        for k, t in st["filter"].items():
            getattr(agent.norm, k).copy_(t)
    if "clipping_decay_schedule" in st:
        agent.ap.algorithm.clipping_decay_schedule = pickle.loads(st["clipping_decay_schedule"])
    random.setstate(st["host_rng"][0])
    np.random.set_state(st["host_rng"][1])
    torch.cuda.synchronize() if torch.cuda.is_available() else None


def _rank_dir(agent, checkpoint_dir):
    """Replay shard, env shard and host RNG state are per rank: under data parallelism every rank
    writes (and restores) its own `rank<r>/` sub-directory; a single process uses the directory itself."""
    dist = getattr(agent, "dist", None)
    if dist is not None and getattr(dist, "world_size", 1) > 1:
        return os.path.join(checkpoint_dir, "rank%d" % dist.rank)
    return checkpoint_dir


def save_checkpoint(agent, checkpoint_dir, checkpoint_id=0):
    """graph_manager.save_checkpoint (:616-637): `<id>_Step-<total_steps>.ckpt` + state file; both are
    published atomically (write to a temporary name, then os.replace)."""
    checkpoint_dir = _rank_dir(agent, checkpoint_dir)
    os.makedirs(checkpoint_dir, exist_ok=True)
    name = "{}_Step-{}.ckpt".format(checkpoint_id, agent.total_steps_counter)
    tmp = os.path.join(checkpoint_dir, name + ".tmp")
    torch.save(agent_state(agent), tmp)
    os.replace(tmp, os.path.join(checkpoint_dir, name))
    tmp = os.path.join(checkpoint_dir, STATE_FILE + ".tmp")
    with open(tmp, "w") as f:
        f.write(name)
    os.replace(tmp, os.path.join(checkpoint_dir, STATE_FILE))
    return name


def latest_checkpoint(checkpoint_dir):
    """checkpoint.py get_checkpoint_state: the state file if present, else the highest id on disk."""
    sf = os.path.join(checkpoint_dir, STATE_FILE)
    if os.path.exists(sf):
        name = open(sf).read().strip()
        if os.path.exists(os.path.join(checkpoint_dir, name)):
            return name
    found = sorted((int(m.group(1)), int(m.group(2)), f) for f in os.listdir(checkpoint_dir)
                   for m in [_NAME.match(f)] if m)
    return found[-1][2] if found else None


def restore_checkpoint(agent, checkpoint_dir, name=None):
    """Checkpoints are trusted input (they are pickles, like the reference's own filter / schedule
    pickles): only restore files this engine wrote."""
    checkpoint_dir = _rank_dir(agent, checkpoint_dir)
    name = name or latest_checkpoint(checkpoint_dir)
    if name is None:
        raise ValueError("No checkpoint to restore in: {}".format(checkpoint_dir))   # graph_manager.py:577-579
    st = torch.load(os.path.join(checkpoint_dir, name), map_location="cpu", weights_only=False)
    load_agent_state(agent, st)
    return name


class NetworkSaver(object):
    """What Architecture.collect_savers hands to the graph manager (saver.py Saver): saves / restores
    ONE device network (weights, target weights, Adam slots) at `<checkpoint prefix>.<path>.pt`."""

    def __init__(self, path, net):
        self.path, self.net = path, net

    def _file(self, save_path):
        return "{}.{}.pt".format(save_path, self.path)

    def save(self, sess, save_path):
        net = self.net
        torch.save({"weights": net.params.weights.cpu(), "target": None if net.target is None else net.target.cpu(),
                    "adam_m": net.adam.m.cpu(), "adam_v": net.adam.v.cpu(), "adam_state": net.adam.state.cpu()},
                   self._file(save_path))
        return [self._file(save_path)]

    def restore(self, sess, restore_path):
        st = torch.load(self._file(restore_path), map_location="cpu", weights_only=False)
        net = self.net
        net.params.weights.copy_(st["weights"])
        if net.target is not None and st["target"] is not None:
            net.target.copy_(st["target"])
        net.adam.m.copy_(st["adam_m"]); net.adam.v.copy_(st["adam_v"]); net.adam.state.copy_(st["adam_state"])
