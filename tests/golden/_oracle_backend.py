"""CPU stand-ins for rl_coach's NetworkWrapper / Architecture objects whose arithmetic is the repo's
numpy oracle (oracle/ac_nets.py, oracle/agents.py).  make_golden.py hands them to the REFERENCE's own
agent classes, so that `DDPGAgent.learn_from_batch`, `TD3Agent.learn_from_batch`,
`SoftActorCriticAgent.learn_from_batch` and `DQNAgent.learn_from_batch` — the reference's code, executed
here — drive the oracle networks.  The resulting weights are stored as fixtures and compared with the
oracle's own whole-update functions (`ddpg_update`, `td3_update`, `sac_update`,
`DQNOracle.learn_from_batch`): that pins the call order, target arithmetic and gradient sign
conventions of the restated updates to the reference.  Build-container only (needs /root/reference)."""
import numpy as np

F32 = np.float32


def _obs(inputs):
    return np.asarray(inputs['observation'], dtype=F32)


class _Wrapper(object):
    has_global = False

    def parallel_prediction(self, pairs):
        return tuple(net.predict(inputs) for net, inputs in pairs)


# ------------------------------------------------------------------------------- DDPG / TD3
class ActorWrapper(_Wrapper):
    def __init__(self, actor):
        self.o = actor
        w = self

        class Online(object):
            gradients_weights_ph = ["gw0"]
            weighted_gradients = ["wg0"]

            def predict(self, inputs, outputs=None, initial_feed_dict=None):
                a = w.o.forward(_obs(inputs))
                if outputs is None:
                    return a
                assert outputs == "wg0"
                w.o.backward(np.asarray(initial_feed_dict["gw0"], dtype=F32))
                return "grads-of-last-backward"

        class Target(object):
            def predict(self, inputs):
                return w.o.forward(_obs(inputs), target=True)

        self.online_network, self.target_network = Online(), Target()

    def apply_gradients_to_online_network(self, gradients, additional_inputs=None):
        assert gradients == "grads-of-last-backward"
        self.o.apply()


class CriticWrapper(_Wrapper):
    def __init__(self, critic, n_act):
        self.o, self.A = critic, n_act
        w = self
        T = critic.T

        def outputs_of(q):
            if T == 1:
                return [q[0][:, None], q[0].mean()]
            return [q[0][:, None], q[1][:, None], np.minimum(q[0], q[1])[:, None], q[0].mean()]

        class Online(object):
            gradients_wrt_inputs = [{"action": "dQ%d/da" % i} for i in range(2 if T == 1 else 4)]

            def predict(self, inputs, outputs=None):
                q = w.o.forward(_obs(inputs), inputs['action'])
                if outputs is None:
                    return outputs_of(q)
                assert outputs == self.gradients_wrt_inputs[-1]["action"]      # the mean-Q output
                return w.o.action_gradient(w.A)

        class Target(object):
            def predict(self, inputs):
                return outputs_of(w.o.forward(_obs(inputs), inputs['action'], target=True))

        self.online_network, self.target_network = Online(), Target()

    def train_and_sync_networks(self, inputs, targets, use_inputs_for_apply_gradients=False):
        self.o.forward(_obs(inputs), inputs['action'])
        losses = self.o.train_backward(np.asarray(targets, dtype=F32))
        norm = self.o.global_norm()
        self.o.apply()
        return float(sum(losses)), [float(x) for x in losses], norm


# -------------------------------------------------------------------------------------- SAC
class SACPolicyWrapper(_Wrapper):
    def __init__(self, pol, n_act):
        w = self
        self.o, self.A, self.normals = pol, n_act, []

        def grads():
            return [g for _, _, l in w.o.layers for g in (l.dW.copy(), l.db.copy())]

        class Head(object):
            policy_mean, actions = "mean", "actions"

        class Online(object):
            gradients_weights_ph = ["gw%d" % i for i in range(6)]
            weighted_gradients = ["wg%d" % i for i in range(6)]
            output_heads = [Head()]

            def predict(self, inputs, outputs=None, initial_feed_dict=None):
                s = _obs(inputs)
                z = np.random.standard_normal((s.shape[0], w.A))       # the graph's sampling op
                w.normals.append(z)
                o = w.o.forward(s, z)
                if outputs == ["mean", "actions"]:                     # choose_action (:312-313)
                    return [o["mean"], o["actions"]]
                if outputs is None:
                    return [o["mean"], o["log_std"], o["raw_actions"], o["actions"], o["logprob"],
                            o["logprob"].mean()]
                if outputs == "wg5":
                    w.o.backward(logp_weight=float(initial_feed_dict["gw5"]))
                else:
                    assert outputs == "wg3"
                    w.o.backward(action_weights=np.asarray(initial_feed_dict["gw3"], dtype=F32))
                return grads()

            def apply_gradients(self, gradients):
                it = iter(gradients)
                for _, _, l in w.o.layers:
                    l.dW, l.db = np.asarray(next(it), dtype=F32), np.asarray(next(it), dtype=F32)
                w.o.apply()

        self.online_network = Online()


class SACQWrapper(_Wrapper):
    def __init__(self, q):
        w = self
        self.o = q

        class Head(object):
            q1_output, q2_output, q1_loss, q2_loss = "q1", "q2", "l1", "l2"

        class Online(object):
            output_heads = [Head()]
            gradients_wrt_inputs = [{"output_0_0": "dq/da"}, {"output_0_0": "dqmean/da"}]

            def predict(self, inputs, outputs=None):
                qv = w.o.forward(_obs(inputs), np.asarray(inputs['output_0_0'], dtype=F32))
                if outputs is None:
                    m = np.minimum(qv[0], qv[1])
                    return [m[:, None], m.mean()]
                if outputs == "dqmean/da":
                    return w.o.action_gradient(qv)
                assert outputs == ["q1", "q2"]
                return qv[0][:, None], qv[1][:, None]

            def train_on_batch(self, inputs, targets, additional_fetches=None):
                y = np.asarray(targets, dtype=F32)[:, 0]
                losses, norm = w.o.train(_obs(inputs), np.asarray(inputs['output_0_0'], dtype=F32), y)
                assert additional_fetches == ["l1", "l2"]
                return float(sum(losses)), [float(sum(losses))], norm, [losses[0], losses[1]]

        self.online_network = Online()


class SACValueWrapper(_Wrapper):
    def __init__(self, v):
        w = self
        self.o = v

        class Online(object):
            def train_on_batch(self, inputs, targets):
                loss = w.o.train(_obs(inputs), np.asarray(targets, dtype=F32)[:, 0])
                return float(loss), [float(loss)], 0.0

        class Target(object):
            def predict(self, inputs):
                return w.o.forward(_obs(inputs), target=True)[:, None]

        self.online_network, self.target_network = Online(), Target()


# -------------------------------------------------------------------------------------- DQN
class DQNWrapper(_Wrapper):
    def __init__(self, dqn_oracle):
        from oracle import losses as L
        w = self
        self.o, self.L = dqn_oracle, L

        class Net(object):
            def __init__(self, target):
                self.target = target

            def predict(self, inputs, outputs=None):
                return w.o.q(_obs(inputs), target=self.target)

        self.online_network, self.target_network = Net(False), Net(True)

    def train_and_sync_networks(self, inputs, targets, importance_weights=None):
        o = self.o
        q = o.q(_obs(inputs))
        wts = None if importance_weights is None else np.asarray(importance_weights, dtype=F32)
        loss, dq = self.L.regression_head_loss(q, np.asarray(targets, dtype=F32), wts, "huber" if o.huber else "mse")
        o.tower.backward(o.head.backward(dq))
        norm = o.global_norm()
        o.adam_step(1.0)
        return float(loss), [float(loss)], norm


# ------------------------------------------------------------------------------ Clipped PPO
class PPOWrapper(_Wrapper):
    """networks['main'] of ClippedPPOAgent on oracle.agents.ClippedPPOOracle (discrete PPOHead)."""

    def __init__(self, net):
        w = self
        self.o, self.frozen = net, None

        class Head(object):
            kl_divergence, entropy = "kl", "entropy"
            likelihood_ratio, clipped_likelihood_ratio = "ratio", "clipped"

        class Online(object):
            output_heads = [object(), Head()]

            def predict(self, inputs, outputs=None):
                s = _obs(inputs)
                return [w.o.values(s)[:, None], w.o.policy_probs(s)]

            def reset_internal_memory(self):
                pass

        class Target(object):
            def predict(self, inputs):
                s = _obs(inputs)
                return [np.zeros((s.shape[0], 1), dtype=F32), w.o.policy_probs(s, w.frozen)]

        self.online_network, self.target_network = Online(), Target()

    def set_is_training(self, state):
        pass

    def sync(self):
        self.frozen = self.o.clone_policy()                     # target <- online

    def train_and_sync_networks(self, inputs, targets, additional_fetches=[]):
        s = _obs(inputs)
        value_targets, advantages = targets
        rescaler = float(inputs['output_1_2'])
        r = self.o.train_minibatch(s, np.asarray(inputs['output_1_0']), np.asarray(advantages, dtype=F32),
                                   np.asarray(value_targets, dtype=F32)[:, 0],
                                   np.asarray(inputs['output_1_1'], dtype=F32), rescaler)
        table = {"kl": r["kl"], "entropy": r["entropy"], "ratio": r["ratio"], "clipped": r["clipped"]}
        return (float(r["value_loss"]) + float(r["total"]), [float(r["value_loss"]), float(r["total"])], r["norm"],
                [table[f] for f in additional_fetches])


class PPOContinuousWrapper(PPOWrapper):
    """networks['main'] of ClippedPPOAgent with a BoxActionSpace: the head's outputs are [policy_mean, policy_std], so
    the old policy travels as TWO inputs (output_1_1, output_1_2) and the clip rescaler moves to output_1_3
    (clipped_ppo_agent.py:258-268)."""

    def __init__(self, net):
        PPOWrapper.__init__(self, net)
        w = self

        class Head(object):
            kl_divergence, entropy = "kl", "entropy"
            likelihood_ratio, clipped_likelihood_ratio = "ratio", "clipped"

        class Online(object):
            output_heads = [object(), Head()]

            def predict(self, inputs, outputs=None):
                s = _obs(inputs)
                mean, std = w.o.policy_mean_std(s)
                return [w.o.values(s)[:, None], mean, std]

            def reset_internal_memory(self):
                pass

        class Target(object):
            def predict(self, inputs):
                s = _obs(inputs)
                mean, std = w.o.policy_mean_std(s, w.frozen)
                return [np.zeros((s.shape[0], 1), dtype=F32), mean, std]

        self.online_network, self.target_network = Online(), Target()

    def sync(self):
        self.frozen = self.o.clone_policy_continuous()

    def train_and_sync_networks(self, inputs, targets, additional_fetches=[]):
        s = _obs(inputs)
        value_targets, advantages = targets
        rescaler = float(inputs['output_1_3'])
        r = self.o.train_minibatch(s, np.asarray(inputs['output_1_0'], dtype=F32), np.asarray(advantages, dtype=F32),
                                   np.asarray(value_targets, dtype=F32)[:, 0],
                                   (np.asarray(inputs['output_1_1'], dtype=F32), np.asarray(inputs['output_1_2'], dtype=F32)),
                                   rescaler)
        table = {"kl": r["kl"], "entropy": r["entropy"], "ratio": r["ratio"], "clipped": r["clipped"]}
        return (float(r["value_loss"]) + float(r["total"]), [float(r["value_loss"]), float(r["total"])], r["norm"],
                [table[f] for f in additional_fetches])
