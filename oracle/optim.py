"""Optimiser and target mixing — numpy fp32 restatement.

tf.train.AdamOptimizer (rl_coach/architectures/tensorflow_components/general_network.py:390-394) is
TensorFlow 1.x code that is not vendored: restated from TF 1.14's published ApplyAdam kernel
(training_ops.cc) — PARITY UNPINNED.  set_weights mixing follows
rl_coach/architectures/tensorflow_components/architecture.py:598-607 exactly (numpy fp32)."""
import numpy as np


class AdamTF1:
    def __init__(self, n, lr, beta1=0.9, beta2=0.99, eps=1e-4):
        self.lr, self.beta1, self.beta2, self.eps = (np.float32(lr), np.float32(beta1),
                                                     np.float32(beta2), np.float32(eps))
        self.m = np.zeros(n, dtype=np.float32)
        self.v = np.zeros(n, dtype=np.float32)
        self.b1p, self.b2p = self.beta1, self.beta2

    def step(self, w, g, grad_scale=1.0):
        one = np.float32(1)
        g = g.astype(np.float32) * np.float32(grad_scale)
        alpha = self.lr * np.sqrt(one - self.b2p) / (one - self.b1p)
        self.m += (g - self.m) * (one - self.beta1)
        self.v += (g * g - self.v) * (one - self.beta2)
        w -= (self.m * alpha) / (np.sqrt(self.v) + self.eps)
        self.b1p = self.b1p * self.beta1
        self.b2p = self.b2p * self.beta2
        return w


def mix_weights(target, online, rate):
    """new_rate * new_weight + (1 - new_rate) * old_weights on fp32 arrays (:604-605)."""
    return rate * online + (1 - rate) * target


def global_norm(g):
    return np.sqrt(np.sum(g.astype(np.float32) ** 2, dtype=np.float32))
