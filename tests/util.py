"""Small helpers shared by the parity tests."""
import ctypes

import numpy as np


def dev_tensor(x, dev, dtype=None):
    import torch
    a = np.ascontiguousarray(np.asarray(x) if dtype is None else np.asarray(x).astype(dtype))
    return torch.from_numpy(a).to(dev)


def status_tensor(dev):
    import torch
    return torch.zeros(1, dtype=torch.int32, device=dev)


class Column(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("row_bytes", ctypes.c_longlong)]


def columns(pairs):
    """pairs: [(src_tensor, dst_tensor)], rows along dim 0."""
    arr = (Column * len(pairs))()
    for i, (s, d) in enumerate(pairs):
        rb = s[0].numel() * s.element_size()
        assert rb == d[0].numel() * d.element_size()
        arr[i] = Column(s.data_ptr(), d.data_ptr(), rb)
    return arr
