"""Data-parallel path on CPU: two gloo ranks (world_size 2) exercise GradientSync exactly as
bench.py / the agents use it — env sharding by rank, identical initial weights, ONE all-reduce of
the flat gradient buffer, the reference's 1/num_workers scaling, identical Adam step on every rank
(weights stay bit-identical without a broadcast), barrier + max-over-ranks timing."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from coach_amd.distributed import GradientSync
    from coach_amd.nn import graph as G
    from oracle.optim import AdamTF1
    from oracle.synth_env import SynthVecEnv
    sync = GradientSync(backend="gloo")
    assert sync.enabled and sync.world_size == world and sync.rank == rank
    # identical initial weights on every rank (same seed), env streams sharded by rank
    params = G.FlatParams()
    d1 = G.Dense(params, "l1", 6, 5, "relu")
    d2 = G.Dense(params, "l2", 5, 2, None)
    params.finalize("cpu")
    rng = np.random.RandomState(0)
    d1.initialize(rng); d2.initialize(rng)
    n_env = 4
    env = SynthVecEnv(1, n_env, 6, 8, 1234, env_id0=rank * n_env)
    obs = env.reset()
    # a rank-local "gradient" that depends on the rank's own observations
    params.grads.copy_(torch.from_numpy(np.resize(obs.astype(np.float32).ravel(), params.size)))
    local = params.grads.clone()
    sync.all_reduce_sum(params.grads)
    scale = sync.grad_scale(True)
    assert scale == 1.0 / world and sync.grad_scale(False) == 1.0
    w = params.weights.numpy().copy()
    adam = AdamTF1(params.size, 1e-3, 0.9, 0.99, 1e-4)
    adam.step(w, params.grads.numpy(), scale)
    sync.barrier()
    t = sync.max_over_ranks(float(rank + 1))
    total = sync.sum_over_ranks(1.0)
    assert sync.min_over_ranks(float(rank + 1)) == 1.0
    # gloo collectives cannot be recorded into a device graph: every rank gets the same "no" (the updates then run as
    # graph segments with eager collectives in between); no all-reduce timing without RCCL
    assert sync.capturable() is False and sync.backend() == "gloo" and sync.all_reduce_us(16) is None
    # the probe's arithmetic against REAL collectives: what PROBE_REPLAYS replays of an in-place sum of ones must leave
    # (round 4 expected W ** 2 after ONE replay and therefore answered "no" on every RCCL job with W >= 2)
    x = torch.ones(8)
    for _ in range(GradientSync.PROBE_REPLAYS):
        sync.all_reduce_sum(x)
    assert GradientSync.PROBE_REPLAYS == 2 and float(x[0]) == GradientSync.probe_expected(world, 2) == float(world) ** 2
    assert GradientSync.probe_expected(world, 1) == float(world) and GradientSync.probe_expected(1, 2) == 1.0
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), w=w, local=local.numpy(), summed=params.grads.numpy(),
             obs=obs, t=t, total=total)
    torch.distributed.destroy_process_group()


def test_gloo_world_size_2(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "r0.npz")
    r1 = np.load(tmp_path / "r1.npz")
    assert not np.array_equal(r0["obs"], r1["obs"])                       # envs sharded by rank
    np.testing.assert_array_equal(r0["summed"], r0["local"] + r1["local"])  # one flat all-reduce
    np.testing.assert_array_equal(r0["summed"], r1["summed"])
    np.testing.assert_array_equal(r0["w"], r1["w"])                       # weights stay bit-identical
    assert float(r0["t"]) == float(r1["t"]) == 2.0 and float(r0["total"]) == 2.0


def test_single_process_sync_is_a_noop():
    import torch
    os.environ.pop("WORLD_SIZE", None)
    os.environ.pop("RANK", None)
    from coach_amd.distributed import GradientSync
    s = GradientSync()
    assert not s.enabled and s.grad_scale(True) == 1.0
    g = torch.ones(8)
    assert s.all_reduce_sum(g) is g and s.max_over_ranks(3.5) == 3.5


def test_stager_ring_cpu():
    import torch
    from coach_amd.staging import Stager, StagerCache
    st = Stager((4,), torch.int32, "cpu")
    for i in range(40):
        d = st.push(np.arange(4, dtype=np.int32) + i)
        assert d is st.dst and d.tolist() == [i, i + 1, i + 2, i + 3]
    c = StagerCache("cpu")
    a = c.push("k", np.ones((2, 3)), torch.float64)
    assert a.shape == (2, 3) and a.dtype == torch.float64


def _worker_async(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from coach_amd.distributed import GradientSync
    sync = GradientSync(backend="gloo")
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    off = 300
    w1 = sync.all_reduce_sum_async(g[off:])          # "late" bucket first, while the rest is "computed"
    g[:off] += 5.0
    w2 = sync.all_reduce_sum_async(g[:off])
    w1.wait(); w2.wait()
    np.save(os.path.join(out_dir, "a%d.npy" % rank), g.numpy())
    torch.distributed.destroy_process_group()


def test_bucketed_async_all_reduce_gloo(tmp_path):
    """The overlap path of ClippedPPOAgent.train_network: two async all-reduces over disjoint slices
    of the flat gradient buffer, waited on before the optimizer step."""
    import torch.multiprocessing as mp
    mp.spawn(_worker_async, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a0, a1 = np.load(tmp_path / "a0.npy"), np.load(tmp_path / "a1.npy")
    base = np.arange(1000, dtype=np.float32)
    exp = base * 3
    exp[:300] += 10.0
    np.testing.assert_array_equal(a0, exp)
    np.testing.assert_array_equal(a1, exp)
