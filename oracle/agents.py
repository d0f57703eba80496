"""Agent update steps on the CPU — numpy restatement of the reference agents' learn_from_batch /
train_network around the oracle network (oracle/nn.py).  Used by the parity tests and timed by
bench.py's cpu_baseline leg.

Follows rl_coach/agents/clipped_ppo_agent.py:209-308 (train_network minibatch), :157-207
(fill_advantages, via oracle.returns) and rl_coach/agents/dqn_agent.py:81-113.
Network numerics are PARITY UNPINNED (TensorFlow absent), see oracle/nn.py.
"""
import numpy as np

from . import losses as L
from . import nn as N
from . import targets as T

F32 = np.float32


class _Base:
    def _all_layers(self):
        out = []
        for prefix, tower, chain in self.chains:
            for i, l in enumerate(chain.layers):
                out.append((self.names[(prefix, tower)][i], tower, l))
        return out

    def adam_step(self, grad_scale=1.0):
        for name, tower, l in self._all_layers():
            self.adam.step((name, tower, "k"), l.W, l.dW, grad_scale)
            self.adam.step((name, tower, "b"), l.b, l.db, grad_scale)

    def grads(self):
        """{name/kernel|bias: {tower: grad}} with the HIP networks' parameter names."""
        out = {}
        for name, tower, l in self._all_layers():
            out.setdefault(name + "/kernel", {})[tower] = l.dW
            out.setdefault(name + "/bias", {})[tower] = l.db
        return out

    def weights(self):
        out = {}
        for name, tower, l in self._all_layers():
            out.setdefault(name + "/kernel", {})[tower] = l.W
            out.setdefault(name + "/bias", {})[tower] = l.b
        return out

    def global_norm(self):
        s = 0.0
        for _, _, l in self._all_layers():
            s += float(np.sum(l.dW.astype(np.float64) ** 2) + np.sum(l.db.astype(np.float64) ** 2))
        return np.sqrt(s)


def _chain_names(arrays, prefix):
    names = []
    i = 0
    while "%s/embedder/conv%d/kernel" % (prefix, i) in arrays:
        names.append("%s/embedder/conv%d" % (prefix, i)); i += 1
    for part in ("embedder", "middleware"):
        i = 0
        while "%s/%s/dense%d/kernel" % (prefix, part, i) in arrays:
            names.append("%s/%s/dense%d" % (prefix, part, i)); i += 1
    return names


class ClippedPPOOracle(_Base):
    """Two separate towers (value, policy) + VHead + discrete PPOHead (clipped_ppo_agent.py:41-58)."""

    def __init__(self, arrays, obs_shape, n_actions, activation="tanh", lr=2.5e-4, beta1=0.9,
                 beta2=0.99, eps=1e-4, clip_eps=0.2, beta_entropy=0.01):
        self.image = len(obs_shape) == 3
        self.A, self.clip_eps, self.beta = n_actions, clip_eps, beta_entropy
        cp = lambda a: {k: [x.copy() for x in v] for k, v in a.items()}
        arrays = cp(arrays)
        self.v_tower = N.build_chain(arrays, "main", 0, obs_shape, activation)
        self.p_tower = N.build_chain(arrays, "main", 1, obs_shape, activation)
        self.v_head = N.Dense(arrays["main/v_head/dense/kernel"][0], arrays["main/v_head/dense/bias"][0])
        self.p_head = N.Dense(arrays["main/ppo_head/policy_fc/kernel"][0],
                              arrays["main/ppo_head/policy_fc/bias"][0])
        tn = _chain_names(arrays, "main")
        self.chains = [("main", 0, self.v_tower), ("main", 1, self.p_tower),
                       ("vh", 0, N.Chain([self.v_head])), ("ph", 0, N.Chain([self.p_head]))]
        self.names = {("main", 0): tn, ("main", 1): tn, ("vh", 0): ["main/v_head/dense"],
                      ("ph", 0): ["main/ppo_head/policy_fc"]}
        self.adam = N.PerTensorAdam(lr, beta1, beta2, eps)

    def clone_policy(self):
        """Frozen copy of (policy tower, head) = the target network's 'old policy' (:238-241)."""
        import copy
        return copy.deepcopy((self.p_tower, self.p_head))

    def policy_probs(self, obs, frozen=None):
        tower, head = frozen if frozen is not None else (self.p_tower, self.p_head)
        return N.softmax(head.forward(tower.forward(N.prep_obs(obs, self.image))))

    def values(self, obs):
        return self.v_head.forward(self.v_tower.forward(N.prep_obs(obs, self.image)))[:, 0]

    def train_minibatch(self, obs, actions, advantages, value_targets, old_probs, clip_rescaler=1.0,
                        grad_scale=1.0):
        x = N.prep_obs(obs, self.image)
        v = self.v_head.forward(self.v_tower.forward(x))
        logits = self.p_head.forward(self.p_tower.forward(x))
        vloss, dv = L.regression_head_loss(v, np.asarray(value_targets, dtype=F32).reshape(-1, 1), None, "mse")
        pl = L.ppo_discrete_loss(logits, actions, advantages, old_probs, self.clip_eps * clip_rescaler, self.beta)
        self.v_tower.backward(self.v_head.backward(dv))
        self.p_tower.backward(self.p_head.backward(pl["dlogits"]))
        norm = self.global_norm()
        self.adam_step(grad_scale)
        return dict(value_loss=vloss, norm=norm, **pl)


class DQNOracle(_Base):
    def __init__(self, arrays, obs_shape, n_actions, activation="relu", lr=2.5e-4, beta1=0.9,
                 beta2=0.99, eps=1e-4, huber=True):
        import copy
        self.image = len(obs_shape) == 3
        self.A, self.huber = n_actions, huber
        arrays = {k: [x.copy() for x in v] for k, v in arrays.items()}
        self.tower = N.build_chain(arrays, "main", 0, obs_shape, activation)
        self.head = N.Dense(arrays["main/q_head/dense/kernel"][0], arrays["main/q_head/dense/bias"][0])
        self.chains = [("main", 0, self.tower), ("qh", 0, N.Chain([self.head]))]
        self.names = {("main", 0): _chain_names(arrays, "main"), ("qh", 0): ["main/q_head/dense"]}
        self.adam = N.PerTensorAdam(lr, beta1, beta2, eps)
        self.target = copy.deepcopy((self.tower, self.head))

    def q(self, obs, target=False):
        tower, head = self.target if target else (self.tower, self.head)
        return head.forward(tower.forward(N.prep_obs(obs, self.image)))

    def update_target(self, rate=1.0):
        from .optim import mix_weights
        for (lt, lo) in zip(self.target[0].layers + [self.target[1]], self.tower.layers + [self.head]):
            lt.W[...] = mix_weights(lt.W, lo.W, F32(rate))
            lt.b[...] = mix_weights(lt.b, lo.b, F32(rate))

    def learn_from_batch(self, obs, next_obs, actions, rewards, game_overs, discount, weights=None,
                         double_dqn=False, grad_scale=1.0):
        q_next = self.q(next_obs, target=True)
        q_next_o = self.q(next_obs) if double_dqn else None
        q = self.q(obs)
        td_targets, td_errors = T.dqn_targets(q_next, q, actions, rewards, game_overs, discount, q_next_o)
        loss, dq = L.regression_head_loss(q, td_targets, weights, "huber" if self.huber else "mse")
        self.tower.backward(self.head.backward(dq))
        norm = self.global_norm()
        self.adam_step(grad_scale)
        return dict(loss=loss, td_errors=td_errors, td_targets=td_targets, norm=norm)
