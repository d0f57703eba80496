"""Network arithmetic — numpy fp32 restatement of the TensorFlow backend's layers.

PARITY UNPINNED for TF's op-level rounding (TensorFlow 1.x is not in /root/reference and cannot be
installed).  Topology / semantics follow rl_coach/architectures/tensorflow_components/:
layers.py:108-121 (conv2d VALID NHWC), :168-185 (dense), embedders/embedder.py:107-121 (input /
255, activation after every layer, flatten), general_network.py:228-405.  Cross-checked in
tests/test_nn.py against torch.nn.functional on CPU (an independent implementation).
"""
import numpy as np

F32 = np.float32


def act(x, kind):
    if kind == "relu":
        return np.maximum(x, F32(0))
    if kind == "tanh":
        return np.tanh(x)
    return x


def act_grad(y, kind):
    if kind == "relu":
        return (y > 0).astype(F32)
    if kind == "tanh":
        return F32(1) - y * y
    return np.ones_like(y)


def im2col(x, k, s):
    B, H, W, C = x.shape
    OH, OW = (H - k) // s + 1, (W - k) // s + 1
    cols = np.empty((B, OH, OW, k, k, C), dtype=x.dtype)
    for ky in range(k):
        for kx in range(k):
            cols[:, :, :, ky, kx, :] = x[:, ky:ky + s * OH:s, kx:kx + s * OW:s, :]
    return cols.reshape(B * OH * OW, k * k * C), OH, OW


def col2im(dcols, shape, k, s):
    B, H, W, C = shape
    OH, OW = (H - k) // s + 1, (W - k) // s + 1
    d = dcols.reshape(B, OH, OW, k, k, C)
    dx = np.zeros(shape, dtype=dcols.dtype)
    for ky in range(k):
        for kx in range(k):
            dx[:, ky:ky + s * OH:s, kx:kx + s * OW:s, :] += d[:, :, :, ky, kx, :]
    return dx


class Dense:
    def __init__(self, W, b, activation=None):
        self.W, self.b, self.act = W.astype(F32), b.astype(F32), activation

    def forward(self, x):
        self.x = x
        self.y = act(x @ self.W + self.b, self.act)
        return self.y

    def backward(self, dy):
        dz = dy * act_grad(self.y, self.act)
        self.dW = self.x.T @ dz
        self.db = dz.sum(0)
        return dz @ self.W.T


class BatchNorm:
    """tf.layers.batch_normalization(x, training) + activation after a dense layer (tensorflow_components/layers.py:26-55;
    TF 1.x non-fused path for [B, C]: tf.nn.moments -> population variance, tf.nn.batch_normalization -> x * inv +
    (beta - mean * inv) with inv = rsqrt(var + eps) * gamma; momentum 0.99, epsilon 1e-3; moving averages move by
    m -= (m - batch) * (1 - momentum) when the UPDATE_OPS run).  PARITY UNPINNED for TF's rounding, like the rest of
    this module."""
    MOMENTUM, EPSILON = 0.99, 1e-3

    def __init__(self, gamma, beta, moving_mean, moving_var, activation=None):
        self.gamma, self.beta = gamma.astype(F32), beta.astype(F32)
        self.moving_mean, self.moving_var = moving_mean.astype(F32), moving_var.astype(F32)
        self.act, self.training, self.pending = activation, False, None

    def forward(self, x):
        x = np.asarray(x, dtype=F32)
        self.x = x
        if self.training:
            B = F32(x.shape[0])
            self.mean = (x.sum(0, dtype=F32) / B).astype(F32)
            d = (x - self.mean).astype(F32)
            self.var = ((d * d).sum(0, dtype=F32) / B).astype(F32)
            self.pending = (self.mean, self.var)
        else:
            self.mean, self.var = self.moving_mean, self.moving_var
        self.r = (F32(1) / np.sqrt(self.var + F32(self.EPSILON))).astype(F32)
        inv = (self.r * self.gamma).astype(F32)
        self.y = act((x * inv + (self.beta - self.mean * inv)).astype(F32), self.act)
        return self.y

    def backward(self, dy, weights=True):
        assert self.training, "the reference differentiates inside Agent.train only"
        du = (np.asarray(dy, dtype=F32) * act_grad(self.y, self.act)).astype(F32)
        xh = ((self.x - self.mean) * self.r).astype(F32)
        B = F32(self.x.shape[0])
        sb, sg = du.sum(0, dtype=F32), (du * xh).sum(0, dtype=F32)
        if weights:
            self.dgamma, self.dbeta = sg.astype(F32), sb.astype(F32)
        return ((self.gamma * self.r) * (du - sb / B - xh * (sg / B))).astype(F32)

    def commit(self):
        """the UPDATE_OPS in front of apply_gradients (architecture.py:273-277)."""
        mean, var = self.pending
        k = F32(1.0 - self.MOMENTUM)
        self.moving_mean = (self.moving_mean - (self.moving_mean - mean) * k).astype(F32)
        self.moving_var = (self.moving_var - (self.moving_var - var) * k).astype(F32)


class Conv:
    def __init__(self, W, b, hwc, kernel, stride, activation=None):
        self.W, self.b, self.act = W.astype(F32), b.astype(F32), activation
        self.hwc, self.k, self.s = hwc, kernel, stride

    def forward(self, x):                       # x: [B, H*W*C] flat or [B,H,W,C]
        B = x.shape[0]
        x = x.reshape((B,) + tuple(self.hwc))
        self.xshape = x.shape
        self.cols, OH, OW = im2col(x, self.k, self.s)
        self.y = act(self.cols @ self.W + self.b, self.act)
        self.out_shape = (B, OH * OW * self.W.shape[1])
        return self.y.reshape(self.out_shape)    # flatten (h, w, c)

    def backward(self, dy):
        dz = dy.reshape(self.y.shape) * act_grad(self.y, self.act)
        self.dW = self.cols.T @ dz
        self.db = dz.sum(0)
        return col2im(dz @ self.W.T, self.xshape, self.k, self.s).reshape(self.xshape[0], -1)


class Chain:
    def __init__(self, layers):
        self.layers = layers

    def forward(self, x):
        for l in self.layers:
            x = l.forward(x)
        return x

    def backward(self, dy):
        for l in reversed(self.layers):
            dy = l.backward(dy)
        return dy


class DuelingHead:
    """DuelingQHead (architectures/tensorflow_components/heads/dueling_q_head.py:33-48): state-value
    stream Dense(512, act) -> Dense(1), action-advantage stream Dense(512, act) -> Dense(A),
    q = V + (A - mean_a A).  Exposed as a 4-layer list [v1, a1, v2, a2] for Adam / norm bookkeeping."""

    def __init__(self, arrays, activation="relu", prefix="main/dueling_q_values_head"):
        k, b = prefix + "/fc1/kernel", prefix + "/fc1/bias"
        self.v1 = Dense(arrays[k][0], arrays[b][0], activation)
        self.a1 = Dense(arrays[k][1], arrays[b][1], activation)
        self.v2 = Dense(arrays[prefix + "/state_value/fc2/kernel"][0], arrays[prefix + "/state_value/fc2/bias"][0])
        self.a2 = Dense(arrays[prefix + "/action_advantage/fc2/kernel"][0],
                        arrays[prefix + "/action_advantage/fc2/bias"][0])
        self.layers = [self.v1, self.a1, self.v2, self.a2]

    def forward(self, x):
        v = self.v2.forward(self.v1.forward(x))
        a = self.a2.forward(self.a1.forward(x))
        mean = (a.sum(1, keepdims=True, dtype=F32) / F32(a.shape[1])).astype(F32)
        return (v + (a - mean)).astype(F32)

    def backward(self, dq):
        dq = dq.astype(F32)
        s = dq.sum(1, keepdims=True, dtype=F32)
        dv, da = s, (dq - s / F32(dq.shape[1])).astype(F32)
        return self.v1.backward(self.v2.backward(dv)) + self.a1.backward(self.a2.backward(da))


def build_chain(arrays, prefix, tower, obs_shape, activation, image_convs=((32, 8, 4), (64, 4, 2), (64, 3, 1))):
    """Rebuild one tower of nn.networks.build_torso from {name: [array per tower]}."""
    layers = []
    i = 0
    hwc = tuple(obs_shape) if len(obs_shape) == 3 else None
    while "%s/embedder/conv%d/kernel" % (prefix, i) in arrays:
        f, k, s = image_convs[i]
        n = "%s/embedder/conv%d" % (prefix, i)
        layers.append(Conv(arrays[n + "/kernel"][tower], arrays[n + "/bias"][tower], hwc, k, s, activation))
        hwc = ((hwc[0] - k) // s + 1, (hwc[1] - k) // s + 1, f)
        i += 1
    for part in ("embedder", "middleware"):
        i = 0
        while "%s/%s/dense%d/kernel" % (prefix, part, i) in arrays:
            n = "%s/%s/dense%d" % (prefix, part, i)
            layers.append(Dense(arrays[n + "/kernel"][tower], arrays[n + "/bias"][tower], activation))
            i += 1
    return Chain(layers)


def prep_obs(obs, image):
    """ObservationEmbedder input rescaling: images / 255 (embedder.py:107-108)."""
    obs = np.asarray(obs)
    if image:
        return (obs.astype(F32) / F32(255.0)).reshape(obs.shape[0], -1)
    return obs.astype(F32).reshape(obs.shape[0], -1)


def softmax(z):
    z = z - z.max(-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(-1, keepdims=True)


class PerTensorAdam:
    """tf.train.AdamOptimizer applied tensor by tensor (elementwise, so identical to one flat pass)."""

    def __init__(self, lr, beta1=0.9, beta2=0.99, eps=1e-4):
        from .optim import AdamTF1
        self._mk = lambda n: AdamTF1(n, lr, beta1, beta2, eps)
        self.slots = {}

    def step(self, key, w, g, grad_scale=1.0):
        if key not in self.slots:
            self.slots[key] = self._mk(w.size)
        flat = w.reshape(-1)
        self.slots[key].step(flat, g.reshape(-1), grad_scale)
        return w


def layer_params(chain):
    return [(l, "kernel", "bias") for l in chain.layers]
