"""Per-episode signal statistics — `Signal` (rl_coach/utils.py:162-212) and the `<name>/Mean | Stdev | Max | Min`
columns Agent.update_log writes (agent.py:548-552), kept on the device.

The reference appends every sample to a Python list and calls np.mean / np.std / np.max / np.min when an episode
is logged.  The hot path cannot afford a host copy per update, so each signal is a 5-double record
{count, sum, sum of squares, max, min} in HBM that one small launch per update folds the new samples into
(rlx_signals_accumulate: every signal of the update in ONE launch); `flush()` reads all records with one copy when a
row is logged and clears them.  mean = sum / n, stdev = sqrt(sumsq / n - mean^2) — np.std's population form — in fp64.
"""
import ctypes
import math

import torch

from . import _rlx


class SignalSource(ctypes.Structure):
    """rlx_signal_source (include/rlx.h)."""
    _fields_ = [("values", ctypes.c_void_p), ("n", ctypes.c_int), ("is_f64", ctypes.c_int), ("signal", ctypes.c_int)]


class DeviceSignals(object):
    def __init__(self, names, device):
        self.names = list(names)
        self.index = {n: i for i, n in enumerate(self.names)}
        self.device = device
        self.lib = _rlx.lib()
        self.table = torch.empty(len(self.names), 5, dtype=torch.float64, device=device)
        self.host_values = {}                                   # constants the host knows (learning rate ...)
        self.reset()

    def reset(self):
        self.lib.signals_reset(self.table, len(self.names), _rlx.current_stream())
        self.host_values = {}

    def add_host_sample(self, name, value):
        self.host_values.setdefault(name, []).append(float(value))

    def accumulate(self, samples):
        """samples: {signal name: device tensor (fp32 / fp64, any shape)}; unknown names are ignored."""
        items = [(self.index[k], v) for k, v in samples.items() if k in self.index and v is not None]
        for i in range(0, len(items), 8):
            chunk = items[i:i + 8]
            arr = (SignalSource * len(chunk))()
            for j, (row, t) in enumerate(chunk):
                if t.dtype not in (torch.float32, torch.float64) or not t.is_contiguous():
                    raise ValueError("signals are contiguous fp32 / fp64 device tensors")
                arr[j] = SignalSource(t.data_ptr(), t.numel(), int(t.dtype == torch.float64), row)
            self.lib.signals_accumulate(arr, len(chunk), self.table, len(self.names), _rlx.current_stream())

    def flush(self):
        """-> {'<name>/Mean': .., '/Stdev', '/Max', '/Min'} ('' for a signal without samples, like Signal.get_mean);
        clears the records.  One device->host copy (a sync: call it once per logged episode, not per update)."""
        t = self.table.cpu().numpy()
        host, out = self.host_values, {}
        for i, name in enumerate(self.names):
            n, s, sq, mx, mn = (float(x) for x in t[i])
            if name in host and host[name]:
                v = host[name]
                n, s, sq = n + len(v), s + sum(v), sq + sum(x * x for x in v)
                mx, mn = max(mx, max(v)), min(mn, min(v))
            if n <= 0:
                stats = ("", "", "", "")
            else:
                mean = s / n
                stats = (mean, math.sqrt(max(sq / n - mean * mean, 0.0)), mx, mn)
            for suffix, value in zip(("Mean", "Stdev", "Max", "Min"), stats):
                out["%s/%s" % (name, suffix)] = value
        self.reset()
        return out
