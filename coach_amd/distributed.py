"""Data parallelism: one process per GPU, environments and replay sharded per rank, ONE all-reduce
of the flat fp32 gradient buffer per update over RCCL/xGMI (torch.distributed backend "nccl" is RCCL
on ROCm; "gloo" is used by the CPU tests).

Replaces the reference's parameter-server accumulation (shared accumulators + spin barriers,
rl_coach/architectures/tensorflow_components/architecture.py:222-231,432-521) — there is no
parameter server: every rank applies the identical Adam step to identical weights, so weights stay
bit-identical without a broadcast.  Gradient scaling mirrors
`scale_down_gradients_by_number_of_workers_for_sync_training` (architecture.py:485-488).

Payloads are 1.5-13.5 MB: a single flat buffer keeps this to one collective per
update; RCCL picks its direct/ring algorithm per message size on the xGMI mesh.
"""
import os
import sys

import torch
import torch.distributed as dist


def quiesce_before_capture(pause_s=0.25):
    """Call before a stream capture in a process that has issued EAGER RCCL collectives.  ProcessGroupNCCL's watchdog thread
    polls the completion event of every eager collective it still holds (every ~100 ms, hipEventQuery); a captured
    collective pulls RCCL's internal stream into the capture, and on ROCm 7.2 a query of an event that was recorded on that
    stream — eagerly, before the capture — then fails with hipErrorCapturedEvent in the watchdog thread, which terminates
    the process (seen once in four runs of tests/test_data_parallel_gpu.py::test_rccl_path_world_size_one, where eager
    timing all-reduces run right in front of the epoch's capture).  Waiting for the device and then for one watchdog
    period lets the watchdog retire those collectives first.  No-op without an initialised RCCL process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl" and torch.cuda.is_available():
        import time
        torch.cuda.synchronize()
        time.sleep(pause_s)


class GradientSync(object):
    def __init__(self, backend=None, force=False):
        """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* come from the launcher (torch.distributed.run).
        backend: "nccl" (= RCCL, the default on a GPU) or "gloo"; force: take the collective path even at world size 1
        (exercises RCCL on a 1-GPU box: bench.py --force-dist)."""
        self.rank = int(os.environ.get("RANK", "0"))
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.enabled = self.world_size > 1 or bool(force)
        if self.enabled and not dist.is_initialized():
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world_size)
        self._capturable = None

    def backend(self):
        return dist.get_backend() if self.enabled else None

    def capturable(self):
        """True when a collective of this process group can be recorded into a hipGraph and replayed (RCCL can; gloo
        cannot): the update then stays ONE graph with its all-reduce as a node — no segment boundary, no eager launch
        between two replays.  Probed once with a real capture + replay of a small all-reduce; every rank takes the
        answer all ranks agree on (a MIN over the ranks), so no rank is left waiting in a collective the others
        skipped.  GradientSync.graph_collectives = False switches the probe (and the feature) off."""
        if self._capturable is None:
            ok = 0
            on_rccl = self.enabled and self.graph_collectives and dist.get_backend() == "nccl" and torch.cuda.is_available()
            if on_rccl:
                try:
                    x = torch.ones(1024, dtype=torch.float32, device="cuda")
                    dist.all_reduce(x)                      # communicator set-up stays outside the capture
                    quiesce_before_capture()
                    x.fill_(1.0)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        dist.all_reduce(x)                  # recorded, not executed: x is still all ones
                    for _ in range(self.PROBE_REPLAYS):
                        g.replay()
                    torch.cuda.synchronize()
                    want = self.probe_expected(self.world_size, self.PROBE_REPLAYS)
                    ok = int(float(x[0].item()) == want and float(x[-1].item()) == want)
                    if not ok:
                        self.probe_note = "replayed all-reduce gave %r, expected %r" % (float(x[0].item()), want)
                except Exception as e:                      # capture refused: the segmented path stays in use
                    ok = 0
                    self.probe_note = "capture refused: %s" % (str(e).splitlines() or [type(e).__name__])[0]
            if self.enabled:
                ok = int(self.min_over_ranks(ok))
            if on_rccl and not ok:
                # not silent: an RCCL job that falls back to graph segments with eager collectives says so once
                sys.stderr.write("coach_amd.distributed: rank %d: all-reduce is not graph-resident (%s); "
                                 "updates run as graph segments around eager collectives\n"
                                 % (self.rank, self.probe_note or "another rank's probe failed"))
            self._capturable = bool(ok)
        return self._capturable

    PROBE_REPLAYS = 2
    probe_note = None

    @staticmethod
    def probe_expected(world_size, replays):
        """What an in-place sum all-reduce of ones holds after `replays` replays of its captured graph on `world_size`
        ranks: every replay multiplies by the world size (the capture itself executes nothing).  Two replays tell a
        graph that really re-executes (W ** 2) from one whose node ran once (W) whenever W >= 2."""
        return float(world_size) ** int(replays)

    graph_collectives = True

    def min_over_ranks(self, value):
        if not self.enabled:
            return value
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item())

    def all_reduce_us(self, n_floats, reps=20):
        """Device time of one in-place all-reduce (sum) of n_floats fp32 over the ranks, back to back on the current
        stream: what bench.py reports as `rccl.allreduce_us` (at world size 1: the collective's fixed cost)."""
        if not self.enabled or dist.get_backend() != "nccl":
            return None
        x = torch.zeros(int(n_floats), dtype=torch.float32, device="cuda")
        for _ in range(3):
            dist.all_reduce(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            dist.all_reduce(x)
        e1.record()
        e1.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps

    def all_reduce_sum(self, flat):
        """In-place sum of a flat gradient buffer over all ranks."""
        if self.enabled:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat

    def all_reduce_sum_async(self, flat):
        """Start an in-place sum on RCCL's stream (it first waits for the work already queued on the
        current stream); `.wait()` on the returned handle makes the current stream wait for it.
        Used to overlap the all-reduce of the early-finished gradients with the rest of backward."""
        if not self.enabled:
            return None
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def grad_scale(self, scale_down_by_workers):
        return 1.0 / self.world_size if (self.enabled and scale_down_by_workers) else 1.0

    def barrier(self):
        if self.enabled:
            if dist.get_backend() == "nccl":      # name the device: no guessing from the rank number
                dist.barrier(device_ids=[torch.cuda.current_device()])
            else:
                dist.barrier()

    def max_over_ranks(self, value):
        if not self.enabled:
            return value
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if not self.enabled:
            return value
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())
