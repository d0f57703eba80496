"""Host -> device staging of small per-step payloads (replay indices, RNG draws).

The destination is ONE static device buffer (hipGraph replays read it), the source is a ring of
pinned host buffers: the H2D copy is asynchronous, so the host may run many steps ahead of the GPU
— a pinned slot is only rewritten after the event recorded behind its last copy has completed.
(Rewriting a single pinned buffer while an earlier async copy from it is still queued silently
ships the NEW contents twice.)
"""
import numpy as np
import torch


class Stager(object):
    def __init__(self, shape, dtype, device, depth=16):
        self.device = device
        self.dst = torch.zeros(shape, dtype=dtype, device=device)
        self.cuda = torch.cuda.is_available() and torch.device(device).type == "cuda"
        self.depth = depth if self.cuda else 1
        self.slots, self.events = [], []
        for _ in range(self.depth):
            h = torch.zeros(shape, dtype=dtype)
            self.slots.append(h.pin_memory() if self.cuda else h)
            self.events.append(None)
        # the pinned slots are FILLED through their numpy views: a torch CPU copy_ of more than 32 K elements goes
        # through the intra-op thread pool (one OpenMP team per copy, 128 threads spinning afterwards on a GPU box) —
        # measured: a 105 KB uint8 record per SAC update made every host call of the loop 5-10 x slower
        self.slots_np = [h.numpy().reshape(-1) for h in self.slots]
        self.i = 0

    def push(self, array):
        """copy a numpy array (same shape) to the static device buffer; returns that buffer."""
        i = self.i
        self.i = (i + 1) % self.depth
        ev = self.events[i]
        if ev is not None:
            ev.synchronize()                      # the copy that last used this slot is done
        self.slots_np[i][...] = np.asarray(array).reshape(-1)
        self.dst.copy_(self.slots[i], non_blocking=True)
        if self.cuda:
            if ev is None:
                ev = torch.cuda.Event()
                self.events[i] = ev
            ev.record()
        return self.dst


    # ---- no copy at all: the consuming kernel reads the PINNED slot itself.  torch's pinned allocations are hipHostMalloc
    # memory, mapped into the device's address space at the same address; a 256-byte payload read over the bus inside a
    # kernel that is waiting on memory anyway costs less than the 3.9 us blit node it replaces (and one dependency less in
    # front of the kernel).  The slot is protected by an event recorded BEHIND the consumer: mapped_done().
    def push_mapped(self, array):
        """write the payload into the next pinned slot and return that slot (pass it to ONE kernel launch, then call
        mapped_done())."""
        i = self.i
        self.i = (i + 1) % self.depth
        ev = self.events[i]
        if ev is not None:
            ev.synchronize()                      # the kernel that last read this slot is done
        self.slots_np[i][...] = np.asarray(array).reshape(-1)
        self._mapped = i
        return self.slots[i]

    def mapped_done(self):
        if self.cuda:
            i = self._mapped
            if self.events[i] is None:
                self.events[i] = torch.cuda.Event()
            self.events[i].record()


class RecordStager(object):
    """Several named host arrays shipped with ONE asynchronous copy: the fields live back to back (16-byte aligned) in
    one pinned record and one static device record; `views[name]` is the typed device view a captured graph reads.
    What a training update needs from the host — the sampled rows, the update's random draws — is one record, one blit."""

    def __init__(self, fields, device, depth=16):
        """fields: [(name, shape, torch dtype), ...]"""
        self.layout, off = {}, 0
        for name, shape, dt in fields:
            off = (off + 15) // 16 * 16
            nbytes = int(np.prod(shape)) * torch.empty(0, dtype=dt).element_size()
            self.layout[name] = (off, nbytes, tuple(shape), dt)
            off += nbytes
        total = (off + 15) // 16 * 16
        self.stager = Stager((total,), torch.uint8, device, depth=depth)
        self.host = np.zeros(total, dtype=np.uint8)
        self.views = {n: self.stager.dst[o:o + b].view(dt).view(*shape) for n, (o, b, shape, dt) in self.layout.items()}
        self.host_views = {n: self.host[o:o + b].view(_NP[dt]).reshape(shape) for n, (o, b, shape, dt) in self.layout.items()}

    def push(self, **arrays):
        for name, a in arrays.items():
            self.host_views[name][...] = a
        self.stager.push(self.host)
        return self.views


class StagerCache(object):
    """{key: Stager} created on first use from the pushed array's shape."""

    def __init__(self, device):
        self.device, self._s = device, {}

    def push(self, key, array, dtype):
        a = np.asarray(array)
        st = self._s.get((key, a.shape))
        if st is None:
            st = Stager(a.shape, dtype, self.device)
            self._s[(key, a.shape)] = st
        return st.push(a.astype(_NP[dtype], copy=False))


_NP = {torch.float64: np.float64, torch.float32: np.float32, torch.int32: np.int32, torch.uint8: np.uint8}
