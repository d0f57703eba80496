"""Host side of the fused continuous-control updates (csrc/ac_fused.hip, csrc/sac_fused.hip): descriptors of the
networks' flat buffers for rlx_td3_fused_* / rlx_sac_fused_*, the workspace they share, and the shape checks that decide
whether an agent's update takes the short launch chain or the layer-by-layer one.

Reference: rl_coach/agents/td3_agent.py:148-209, rl_coach/agents/soft_actor_critic_agent.py:168-280 — the update LOGIC
(pass order, targets, which weights a pass sees) is the reference's; what changes is how many launches carry it."""
import ctypes

import torch

from .. import _rlx
from . import graph as G


def _dense_layers(seq):
    layers = seq.layers if hasattr(seq, "layers") else [seq]
    return layers if all(isinstance(l, G.Dense) for l in layers) else None


def mlp3(params, l1, l2, l3):
    """rlx_mlp3 of three Dense layers that live in `params` (tower 0 offsets + the towers' strides)."""
    m = _rlx.Mlp3()
    for i, l in enumerate((l1, l2, l3), 1):
        setattr(m, "off_w%d" % i, params.entries[l.kname][0])
        setattr(m, "off_b%d" % i, params.entries[l.bname][0])
        setattr(m, "tower_stride%d" % i, params.stride(l.kname) if l.T > 1 else 0)
    m.d_in, m.h1, m.h2, m.d_out = l1.K, l1.N, l2.N, l3.N
    return m


def fused_net(net, grad_scale=1.0, mix_rate=None, with_norm=False):
    """rlx_fused_net of a _NetBase: its flat buffers, Adam slots and hyper-parameters."""
    n = _rlx.FusedNet()
    p, ad = net.params, net.adam
    n.weights = p.weights.data_ptr()
    n.target_weights = net.target.data_ptr() if net.target is not None else None
    n.adam_m, n.adam_v, n.adam_state = ad.m.data_ptr(), ad.v.data_ptr(), ad.state.data_ptr()
    n.grads = p.grads.data_ptr()
    n.norm_out = net.norm.data_ptr() if with_norm else None
    n.ticket = ad.ticket.data_ptr()
    n.learning_rate, n.beta1, n.beta2, n.epsilon = ad.lr, ad.beta1, ad.beta2, ad.eps
    n.grad_scale = float(grad_scale)
    n.mix_rate = -1.0 if mix_rate is None or net.target is None else float(mix_rate)
    return n


class FusedTD3(object):
    """TD3Agent's update through rlx_td3_fused_critic_update / rlx_td3_fused_actor_update.  `supported(agent)` is the
    whole rule for taking it: the Mujoco_TD3 topology (actor: one embedder layer + one middleware layer + tanh head;
    twin critic: no embedder, two middleware layers, Dense(1) heads), relu, no batch normalisation."""

    @staticmethod
    def layers(agent):
        actor, critic = agent.networks["actor"], agent.networks["critic"]
        ae, am, ce, cm = (_dense_layers(x) for x in (actor.emb, actor.mid, critic.emb, critic.mid))
        if None in (ae, am, ce, cm) or len(ae) != 1 or len(am) != 1 or len(ce) != 0 or len(cm) != 2:
            return None
        if actor.head_bn is not None or actor.bn_layers or critic.bn_layers or critic.T != 2:
            return None
        if any(l.act != "relu" for l in ae + am + cm) or actor.head.act != "tanh" or critic.head.act is not None:
            return None
        if getattr(agent.ap.algorithm, "action_penalty", 0):
            return None
        return (ae[0], am[0], actor.head), (cm[0], cm[1], critic.head)

    def __init__(self, agent):
        self.agent = agent
        ls = self.layers(agent)
        if ls is None:
            raise ValueError("FusedTD3: unsupported network topology")
        actor, critic = agent.networks["actor"], agent.networks["critic"]
        d = self.desc = _rlx.Td3FusedDesc()
        d.actor_mlp = mlp3(actor.params, *ls[0])
        d.critic_mlp = mlp3(critic.params, *ls[1])
        d.batch, d.obs_dim, d.act_dim = agent.batch_size, agent.obs_dim, agent.A
        d.actor_scale = float(actor._uniform_scale)
        if not _rlx.lib().td3_fused_supported(ctypes.byref(d)):
            raise ValueError("FusedTD3: unsupported shape")
        need = ctypes.c_longlong()
        _rlx.lib().td3_fused_workspace_floats(ctypes.byref(d), ctypes.byref(need))
        self.ws = torch.zeros(need.value, dtype=torch.float32, device=agent.device)
        d.workspace, d.workspace_floats = self.ws.data_ptr(), need.value
        alg = agent.ap.algorithm
        d.noise_clip, d.discount = float(alg.noise_clipping), float(alg.discount)
        clip = alg.clip_critic_targets
        d.has_clip = int(clip is not None)
        d.clip_low, d.clip_high = (float(clip[0]), float(clip[1])) if clip else (0.0, 0.0)
        d.use_non_zero_discount_for_terminal_states = int(bool(alg.use_non_zero_discount_for_terminal_states))
        d.action_low, d.action_high = agent.d_low.data_ptr(), agent.d_high.data_ptr()
        d.td_targets, d.q_min = agent.td_targets.data_ptr(), agent.q_min.data_ptr()
        d.loss, d.neg_action_grad = critic.loss.data_ptr(), agent.neg_action_grad.data_ptr()

    @classmethod
    def supported(cls, agent):
        try:
            cls(agent)
            return True
        except (ValueError, AttributeError):
            return False

    def _fill(self, b, mix):
        a, d = self.agent, self.desc
        actor, critic = a.networks["actor"], a.networks["critic"]
        d.actor = fused_net(actor, a._scale("actor"), mix)
        d.critic = fused_net(critic, a._scale("critic"), mix, with_norm=True)
        d.obs = b._states["observation"].data_ptr()
        d.next_obs = b._next_states["observation"].data_ptr()
        d.actions, d.rewards, d.game_overs = b.actions().data_ptr(), b.rewards().data_ptr(), b.game_overs().data_ptr()
        d.noise = a.noise.data_ptr()

    def critic_update(self, b, mix=None, write_grads=False):
        self._fill(b, mix)
        _rlx.lib().td3_fused_critic_update(ctypes.byref(self.desc), int(write_grads), _rlx.current_stream())

    def actor_update(self, b, mix=None, write_grads=False):
        self._fill(b, mix)
        _rlx.lib().td3_fused_actor_update(ctypes.byref(self.desc), int(write_grads), _rlx.current_stream())


class FusedSAC(object):
    """SoftActorCriticAgent's update through rlx_sac_fused_update (six launches).  Taken for the Mujoco_SAC topology: policy
    and V with one embedder layer + one middleware layer (relu), the twin Q head with two layers of one width."""

    @staticmethod
    def layers(agent):
        pol, q, v = (agent.networks[k] for k in ("policy", "q", "v"))
        pe, pm, ve, vm = (_dense_layers(x) for x in (pol.emb, pol.mid, v.emb, v.mid))
        if None in (pe, pm, ve, vm) or any(len(x) != 1 for x in (pe, pm, ve, vm)) or len(q.fcs) != 1:
            return None
        if any(l.act != "relu" for l in pe + pm + ve + vm + [q.obs_fc, q.act_fc] + q.fcs):
            return None
        if pol.head.act is not None or v.head.act is not None or q.out.act is not None or q.fcs[0].N != q.h0:
            return None
        return (pe[0], pm[0], pol.head), (ve[0], vm[0], v.head)

    def __init__(self, agent):
        self.agent = agent
        ls = self.layers(agent)
        if ls is None:
            raise ValueError("FusedSAC: unsupported network topology")
        pol, q, v = (agent.networks[k] for k in ("policy", "q", "v"))
        d = self.desc = _rlx.SacFusedDesc()
        d.policy_mlp = mlp3(pol.params, *ls[0])
        d.v_mlp = mlp3(v.params, *ls[1])
        P = q.params
        for tag, l in (("obs", q.obs_fc), ("act", q.act_fc), ("fc", q.fcs[0]), ("out", q.out)):
            setattr(d, "q_off_%s_w" % tag, P.entries[l.kname][0])
            setattr(d, "q_off_%s_b" % tag, P.entries[l.bname][0])
            if P.stride(l.kname) != P.stride(l.bname):
                raise ValueError("FusedSAC: kernel and bias tower strides differ")
            setattr(d, "q_stride_%s" % tag, P.stride(l.kname))
        d.batch, d.obs_dim, d.act_dim, d.q_hidden = agent.batch_size, agent.obs_dim, agent.A, q.h0
        if not _rlx.lib().sac_fused_supported(ctypes.byref(d)):
            raise ValueError("FusedSAC: unsupported shape")
        need = ctypes.c_longlong()
        _rlx.lib().sac_fused_workspace_floats(ctypes.byref(d), ctypes.byref(need))
        self.ws = torch.zeros(need.value, dtype=torch.float32, device=agent.device)
        d.workspace, d.workspace_floats = self.ws.data_ptr(), need.value
        alg = agent.ap.algorithm
        d.discount = float(alg.discount)
        d.resample_noise_per_pass = int(bool(alg.resample_noise_per_pass))
        d.value_targets, d.log_target = agent.value_targets.data_ptr(), agent.log_target.data_ptr()
        d.td_targets, d.dq_da = agent.td_targets.data_ptr(), agent.dq_da.data_ptr()
        d.q_loss, d.v_loss = q.loss.data_ptr(), v.loss.data_ptr()

    @classmethod
    def supported(cls, agent):
        try:
            cls(agent)
            return True
        except (ValueError, AttributeError):
            return False

    def update(self, b, mix=None, write_grads=False):
        a, d = self.agent, self.desc
        pol, q, v = (a.networks[k] for k in ("policy", "q", "v"))
        d.policy = fused_net(pol, a._scale("policy"))
        d.q = fused_net(q, a._scale("q"), with_norm=True)
        d.v = fused_net(v, a._scale("v"), mix)
        d.obs = b._states["observation"].data_ptr()
        d.next_obs = b._next_states["observation"].data_ptr()
        d.actions, d.rewards, d.game_overs = b.actions().data_ptr(), b.rewards().data_ptr(), b.game_overs().data_ptr()
        d.normals = a.normals.data_ptr()
        _rlx.lib().sac_fused_update(ctypes.byref(d), int(write_grads), _rlx.current_stream())
