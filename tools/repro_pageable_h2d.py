"""Is an ASYNCHRONOUS host-to-device copy from PAGEABLE memory ordered against the host on this ROCm stack?  (One of the
hypotheses for the "training-graph fault", DESIGN.md section 5: the eager AdamW step used to upload its per-parameter pointer
table with `torch.from_numpy(blob).to(dev, non_blocking=True)` and drop `blob`.)  torch only, no gridmm code.

    python tools/repro_pageable_h2d.py

The device is kept busy, a temporary numpy array is uploaded with `.to(device, non_blocking=True)` and dropped, and the host
immediately reuses the freed pages.  Result on ROCm 7.2 / MI355X (profiles/r4_train_graph_fault.txt): 0 corrupted uploads at
every size from 1 KiB to 4 MiB -- the copy is staged before the call returns, as on CUDA.  Hypothesis refuted."""
import numpy as np
import torch

dev = torch.device("cuda:0")
a = torch.randn(4096, 4096, device=dev)
print("torch", torch.__version__, "hip", torch.version.hip)
for nbytes in (1 << 10, 1 << 12, 1 << 14, 1 << 15, 1 << 16, 1 << 18, 1 << 20, 1 << 22):
    bad = 0
    trials = 40
    for it in range(trials):
        b = a
        for _ in range(30):                       # a few ms of queued device work in front of the copy
            b = (b @ a) * 1e-3
        tag = it % 200 + 1
        blob = np.full(nbytes, tag, np.uint8)     # pageable temporary, like the table built per optimizer step
        d = torch.from_numpy(blob).to(dev, non_blocking=True)
        del blob
        junk = [np.full(nbytes, 255, np.uint8) for _ in range(4)]     # the allocator hands the same pages out again
        torch.cuda.synchronize()
        bad += int((d != tag).any())
        del junk
    print("%8d bytes: %d of %d uploads arrived with bytes written AFTER .to(non_blocking=True) returned" % (nbytes, bad, trials), flush=True)
