"""K5/K6 prioritized replay: oracle vs the reference's golden traces (CPU), HIP vs oracle (GPU).

Mirrors rl_coach/tests/memories/test_prioritized_experience_replay.py (sum/min/max totals,
get_element_by_partial_sum) plus full store/sample/update traces produced by the reference
itself (tests/golden/make_golden.py::gen_per).
"""
import numpy as np
import pytest

from oracle.per import PrioritizedReplayOracle, SegmentTree

CASES = ["c8", "c50", "c1024", "c16k"]


# ------------------------------------------------------------------ reference's own unit vectors
def test_oracle_sum_tree_reference_vectors():
    # rl_coach/tests/memories/test_prioritized_experience_replay.py:13-40
    t = SegmentTree(size=4, op='sum')
    for v in (10, 20, 5, 7.5):
        t.add(v)
    assert t.total_value() == 42.5
    t.add(2.5)                      # wraps: overwrites leaf 0
    t.add(5)
    assert t.total_value() == 20
    assert t.get_element_by_partial_sum(2)[0] == 0
    assert t.get_element_by_partial_sum(3)[0] == 1
    assert t.get_element_by_partial_sum(10)[0] == 2
    assert t.get_element_by_partial_sum(13)[0] == 3
    t.update(2, 10)
    assert t.tree[2 + 3] == 10 and t.total_value() == 25
    with pytest.raises(ValueError):
        SegmentTree(size=5, op='sum')


def test_oracle_min_max_tree_reference_vectors():
    # :43-86 of the same reference test
    t = SegmentTree(size=4, op='min')
    for v in (10, 20, 5, 7.5):
        t.add(v)
    assert t.total_value() == 5
    t.add(2)
    assert t.total_value() == 2
    t.add(3); t.add(3); t.add(3); t.add(5)
    assert t.total_value() == 3
    t = SegmentTree(size=4, op='max')
    for v in (10, 20, 5, 7.5):
        t.add(v)
    assert t.total_value() == 20
    t.add(2)
    assert t.total_value() == 20
    t.add(3); t.add(3); t.add(3); t.add(5)
    assert t.total_value() == 5
    t.update(1, 10)
    assert t.total_value() == 10


def test_oracle_appendix_b(golden):
    g = golden("per")
    m = PrioritizedReplayOracle(8, alpha=0.6, beta=0.4)
    for _ in range(8):
        m.store()
    m.update_priorities(range(8), [.1, .5, 1, 2, 0, .3, 4, .7])
    idx, w = m.sample(4, g["appB_u"])
    assert idx.tolist() == g["appB_idx"].tolist() == [0, 2, 6, 6]
    assert w.tolist() == g["appB_w"].tolist()
    assert [m.sum_tree.total_value(), m.min_tree.total_value(), m.maximal_priority] == \
        g["appB_roots"].tolist()


def test_oracle_double_store_quirk(golden):
    g = golden("per")
    m = PrioritizedReplayOracle(8)
    counts = []
    for _ in range(6):
        m.store()
        counts.append(m.num_transitions())
    assert counts == g["quirk_counts"].tolist()


def _replay(g, name, backend):
    """Re-run the golden trace `name` on `backend` (object with store/update_priorities/sample/
    roots/trees) and compare every recorded quantity."""
    max_size, cap, batch, n0, rounds, spr, seed = g[name + "_meta"].tolist()
    for _ in range(n0):
        backend.store()
    init_err = g[name + "_init_err"]
    backend.update_priorities(np.arange(len(init_err)), init_err)
    for r in range(rounds):
        for _ in range(spr):
            backend.store()
        assert backend.num_transitions() == g[name + "_ntrans"][r]
        idx, w = backend.sample(batch, g[name + "_u"][r])
        backend.check_sample(idx, w, g[name + "_idx"][r], g[name + "_w"][r])
        backend.update_priorities(g[name + "_idx"][r], g[name + "_err"][r])
        backend.check_roots(g[name + "_sum_root"][r], g[name + "_min_root"][r], g[name + "_maxp"][r])
    backend.check_trees(g[name + "_sum_tree"], g[name + "_min_tree"], g[name + "_max_tree"])


class _OracleBackend:
    def __init__(self, max_size, alpha, beta, eps):
        self.m = PrioritizedReplayOracle(max_size, alpha, beta, eps)
        self.store = self.m.store
        self.update_priorities = self.m.update_priorities
        self.sample = self.m.sample
        self.num_transitions = self.m.num_transitions

    def check_sample(self, idx, w, gidx, gw):
        assert idx.tolist() == gidx.tolist()
        assert w.tolist() == gw.tolist()                      # bit-exact fp64

    def check_roots(self, s, mn, mp):
        assert (self.m.sum_tree.total_value(), self.m.min_tree.total_value(),
                self.m.maximal_priority) == (s, mn, mp)

    def check_trees(self, s, mn, mx):
        k = len(s)
        assert np.array_equal(self.m.sum_tree.tree[:k], s)
        assert np.array_equal(self.m.min_tree.tree[:k], mn)
        assert np.array_equal(self.m.max_tree.tree[:k], mx)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_trace(golden, name):
    g = golden("per")
    alpha, beta, eps = g[name + "_ab"].tolist()
    _replay(g, name, _OracleBackend(int(g[name + "_meta"][0]), alpha, beta, eps))


# --------------------------------------------------------------------------------- HIP (C ABI)
class _HipBackend:
    """Drives rlx_per_* directly.  exact=False is the product default: p**alpha on the device with
    rlx::libm_pow (glibc's algorithm and tables, csrc/libm_pow.hpp); exact=True feeds host-computed
    p**alpha leaves (rlx_per_update_leaves).  BOTH must give trees, indices and importance weights
    that are bit-identical to the reference's traces."""

    def __init__(self, rlx, dev, max_size, alpha, beta, eps, exact):
        import torch
        self.torch, self.rlx, self.dev = torch, rlx, dev
        self.cap = 1
        while self.cap < max_size:
            self.cap *= 2
        n = 2 * self.cap - 1
        self.sum = torch.empty(n, dtype=torch.float64, device=dev)
        self.min = torch.empty(n, dtype=torch.float64, device=dev)
        self.max = torch.empty(n, dtype=torch.float64, device=dev)
        self.maxp = torch.zeros(1, dtype=torch.float64, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.alpha, self.beta, self.eps, self.exact = alpha, beta, eps, exact
        self.next_leaf = 0
        self.list_len = 0
        rlx.per_init(self.sum, self.min, self.max, self.cap, self.maxp, 0)

    def num_transitions(self):
        return self.list_len

    def store(self, n=1):
        if self.exact:      # host mirrors maximal_priority -> libm pow, bit-identical leaves
            p = self.maxp.item()
            self.rlx.per_store_value(self.sum, self.min, self.max, self.cap, self.next_leaf, n,
                                     p ** self.alpha, p, self.maxp, self.status, 0)
        else:
            self.rlx.per_store(self.sum, self.min, self.max, self.cap, self.next_leaf, n,
                               self.alpha, self.maxp, self.status, 0)
        self.next_leaf = (self.next_leaf + n) % self.cap
        self.list_len = min(self.list_len + 2 * n, self.cap)       # double-store quirk

    def update_priorities(self, idx, err):
        t = self.torch
        idx_d = t.as_tensor(np.asarray(idx, dtype=np.int32), device=self.dev)
        err = np.asarray(err, dtype=np.float64)
        if self.exact:
            p = err + self.eps
            pa = np.array([float(x) ** self.alpha for x in p])
            self.rlx.per_update_leaves(self.sum, self.min, self.max, self.cap, idx_d,
                                       t.as_tensor(pa, device=self.dev),
                                       t.as_tensor(p, device=self.dev), len(idx), self.maxp,
                                       self.status, 0)
        else:
            self.rlx.per_update(self.sum, self.min, self.max, self.cap, idx_d,
                                t.as_tensor(err, device=self.dev), len(idx), self.alpha, self.eps,
                                self.maxp, self.status, 0)
        assert int(self.status.item()) == 0

    def sample(self, size, u):
        t = self.torch
        idx = t.empty(size, dtype=t.int32, device=self.dev)
        w = t.empty(size, dtype=t.float64, device=self.dev)
        self.rlx.per_sample(self.sum, self.min, self.cap,
                            t.as_tensor(np.asarray(u, dtype=np.float64), device=self.dev), size,
                            float(self.list_len), self.beta, idx, w, None, 0, 0, None, 0)
        return idx.cpu().numpy(), w.cpu().numpy()

    def check_sample(self, idx, w, gidx, gw):
        assert idx.tolist() == gidx.tolist()                  # indices bit-exact
        assert w.tolist() == gw.tolist()                      # importance weights bit-exact (fp64)

    def check_roots(self, s, mn, mp):
        assert (self.sum[0].item(), self.min[0].item(), self.maxp.item()) == (s, mn, mp)

    def check_trees(self, s, mn, mx):
        k = len(s)
        for mine, ref in ((self.sum, s), (self.min, mn), (self.max, mx)):
            assert np.array_equal(mine[:k].cpu().numpy(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_reference_trace(golden, rlx, dev, name, exact):
    g = golden("per")
    alpha, beta, eps = g[name + "_ab"].tolist()
    _replay(g, name, _HipBackend(rlx, dev, int(g[name + "_meta"][0]), alpha, beta, eps, exact))


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("cap_log2,batch", [(3, 4), (10, 32), (17, 64), (20, 32), (20, 1024)])
def test_hip_vs_oracle_random(rlx, dev, cap_log2, batch, exact):
    """Seeded random traces up to the BASELINE capacity 2^20 (C3): bulk vector-stores, updates
    with duplicate indices, stratified samples; indices bit-exact against the oracle."""
    rng = np.random.RandomState(cap_log2 * 100 + batch)
    cap = 1 << cap_log2
    o = PrioritizedReplayOracle(cap, 0.6, 0.4, 1e-6)
    h = _HipBackend(rlx, dev, cap, 0.6, 0.4, 1e-6, exact=exact)
    # bulk fill (vectorised store of n leaves == n sequential stores)
    fill = cap if cap_log2 <= 17 else 3000
    for _ in range(fill):
        o.store()
    done = 0
    while done < fill:
        n = min(1000, fill - done)
        h.store(n)
        done += n
    rounds = 4 if cap_log2 <= 17 else 3
    for r in range(rounds):
        idx = rng.randint(0, min(fill, cap), size=batch)
        if batch >= 4:
            idx[-1] = idx[0]                                   # duplicate: last one wins
        err = np.abs(rng.randn(batch)) * 3
        o.update_priorities(idx.tolist(), err.tolist())
        h.update_priorities(idx, err)
        u = rng.random_sample(batch)
        oi, ow = o.sample(batch, u)
        hi, hw = h.sample(batch, u)
        assert hi.tolist() == oi.tolist()
        assert hw.tolist() == ow.tolist()
        assert h.sum[0].item() == o.sum_tree.total_value()
        assert h.min[0].item() == o.min_tree.total_value()
        assert h.maxp.item() == o.maximal_priority
        for _ in range(7):
            o.store()
        h.store(7)
    k = min(2 * cap - 1, 4095)
    assert np.array_equal(h.sum[:k].cpu().numpy(), o.sum_tree.tree[:k])
    assert np.array_equal(h.max[:k].cpu().numpy(), o.max_tree.tree[:k])


@pytest.mark.gpu
def test_hip_per_error_reporting(rlx, dev):
    import torch
    from coach_amd._rlx import RlxError
    t = torch.empty(9, dtype=torch.float64, device=dev)
    mp = torch.zeros(1, dtype=torch.float64, device=dev)
    with pytest.raises(RlxError, match="power of 2"):
        rlx.per_init(t, t, t, 5, mp, 0)                       # reference: ValueError (:62-63)
    h = _HipBackend(rlx, dev, 8, 0.6, 0.4, 1e-6, exact=False)
    h.store(8)
    idx = torch.tensor([1, 9], dtype=torch.int32, device=dev)        # 9 is out of range
    err = torch.tensor([1.0, -1.0], dtype=torch.float64, device=dev)  # negative error
    rlx.per_update(h.sum, h.min, h.max, h.cap, idx, err, 2, 0.6, 1e-6, h.maxp, h.status, 0)
    assert int(h.status.item()) & 1
    h.status.zero_()
    idx = torch.tensor([1, 2], dtype=torch.int32, device=dev)
    rlx.per_update(h.sum, h.min, h.max, h.cap, idx, err, 2, 0.6, 1e-6, h.maxp, h.status, 0)
    assert int(h.status.item()) & 2


@pytest.mark.gpu
@pytest.mark.parametrize("cap,n,dups", [(1 << 20, 64, 0), (1 << 20, 256, 9), (1 << 12, 200, 40), (64, 64, 30),
                                        (2, 2, 1), (1 << 16, 1, 0), (1 << 16, 7, 6)])
def test_both_priority_update_kernels_write_the_same_trees(rlx, dev, cap, n, dups):
    """rlx_per_tuning: the level-synchronous kernel and the one whose threads meet only in LDS are two schedules of
    the same arithmetic (_propagate, segment_tree.py:63-74) — trees and maximal priority bit-identical, duplicates
    (the last occurrence wins, :214-215) and neighbouring leaves included."""
    import torch
    trees = []
    for path_max in (0, 256):
        rlx.per_tuning(path_max)
        try:
            h = _HipBackend(rlx, dev, cap, 0.6, 0.4, 1e-6, exact=False)
            h.store(min(cap, 4096))
            r = np.random.RandomState(7)
            for _ in range(6):
                idx = r.randint(0, cap, n).astype(np.int32)
                if n > 3:
                    idx[1] = idx[0] ^ 1                       # siblings
                    idx[2] = (idx[0] + 2) % cap               # cousins
                for d in range(dups):
                    idx[r.randint(0, n)] = idx[r.randint(0, n)]
                err = r.rand(n) * 5.0
                rlx.per_update(h.sum, h.min, h.max, h.cap, torch.from_numpy(idx).to(dev),
                               torch.from_numpy(err).to(dev), n, 0.6, 1e-6, h.maxp, h.status, 0)
                # a store of consecutive leaves (the contiguous fast path), one that wraps around the ring, one leaf
                for start, m in ((int(r.randint(0, cap)), min(n, cap)), (cap - 1, min(3, cap)), (int(r.randint(0, cap)), 1)):
                    rlx.per_store(h.sum, h.min, h.max, h.cap, start if start + m <= cap or m > 1 else 0, m, 0.6, h.maxp,
                                  h.status, 0)
            assert int(h.status.item()) == 0
            trees.append([t.cpu().numpy().copy() for t in (h.sum, h.min, h.max, h.maxp)])
        finally:
            rlx.per_tuning(256)
    for a, b in zip(*trees):
        assert np.array_equal(a, b)
