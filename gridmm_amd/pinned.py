"""A small ring of pinned host buffers for asynchronous host-to-device copies.

A pinned source buffer may be rewritten only after the copy issued from it has run.  With ONE buffer the writer waits for the
previous copy -- which sits behind everything queued on the stream before it (in a training loop the whole backward of the
previous iteration).  The ring hands out a slot whose copy has completed; it grows (pinned allocations cost milliseconds, so
only when no slot is free and the ring is below `max_slots`) and otherwise waits for the oldest slot."""
import torch


class PinnedRing:
    def __init__(self, numel, dtype=torch.uint8, max_slots=16, first=None):
        self.numel, self.dtype, self.max_slots = int(numel), dtype, int(max_slots)
        self.slots = [[first if first is not None else torch.zeros(self.numel, dtype=dtype).pin_memory(), None]]
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
            self.slots.append([torch.zeros(self.numel, dtype=self.dtype).pin_memory(), None])
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
