"""A small ring of pinned host buffers for asynchronous host-to-device copies.

A pinned source buffer may be rewritten only after the copy issued from it has run.  With ONE buffer the writer waits for the
previous copy -- which sits behind everything queued on the stream before it (in a training loop the whole backward of the
previous iteration).  The ring hands out a slot whose copy has completed; it grows (pinned allocations cost milliseconds, so
only when no slot is free and the ring is below `max_slots`) and otherwise waits for the oldest slot."""
import torch


def pinned_slots(numel, count, dtype=torch.uint8):
    """`count` pinned buffers of `numel` elements carved from ONE pinned allocation: the cost of pinning is per call (~2 ms
    measured in the fine-tune loop, whatever the size at these sizes), so rings grow by several slots at a time."""
    big = torch.zeros(int(numel) * int(count), dtype=dtype).pin_memory()
    return [big[i * numel:(i + 1) * numel] for i in range(count)]


class PinnedRing:
    GROW = 4          # slots added per pinned allocation

    def __init__(self, numel, dtype=torch.uint8, max_slots=16, first=None):
        self.numel, self.dtype, self.max_slots = int(numel), dtype, int(max_slots)
        self.slots = [[first if first is not None else torch.zeros(self.numel, dtype=dtype).pin_memory(), None]]
        self._spare = []
        self.pos = 0

    def acquire(self):
        """Index of a slot that may be written now."""
        n = len(self.slots)
        for k in range(1, n + 1):
            i = (self.pos + k) % n
            ev = self.slots[i][1]
            if ev is None or ev.query():
                self.pos = i
                return i
        if n < self.max_slots:
            if not self._spare:
                self._spare = pinned_slots(self.numel, min(self.GROW, self.max_slots - n), self.dtype)
            self.slots.append([self._spare.pop(), None])
            self.pos = n
            return n
        self.pos = (self.pos + 1) % n
        self.slots[self.pos][1].synchronize()
        return self.pos

    def host(self, i):
        return self.slots[i][0]

    def record(self, i, event=None):
        """Call right after enqueueing the copy that reads slot i (on the current stream)."""
        if event is None:
            event = torch.cuda.Event()
            event.record()
        self.slots[i][1] = event
        return event


_BLOB_RINGS = {}


def upload_blob(blob, device):
    """A small host table (numpy uint8 array) -> device tensor through a per-size-class pinned ring: one asynchronous copy, no
    pinned allocation per call (a `torch.from_numpy(blob).pin_memory()` is one, ~2 ms each)."""
    n = int(blob.nbytes)
    cls = max(1 << 12, 1 << (n - 1).bit_length())
    ring = _BLOB_RINGS.get((cls, str(device)))
    if ring is None:
        ring = _BLOB_RINGS[(cls, str(device))] = PinnedRing(cls, max_slots=8)
    k = ring.acquire()
    host = ring.host(k)
    host.numpy()[:n] = blob.reshape(-1).view("uint8")
    d = torch.empty(n, dtype=torch.uint8, device=device)
    d.copy_(host[:n], non_blocking=True)
    ring.record(k)
    return d
