"""In-kernel cycle profile of attention_rows (build attention_lds.hip with -DGRIDMM_ATT_PROF and relink):
   cd gridmm_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DGRIDMM_ATT_PROF -c attention_lds.hip -o build/attention_lds.o && make"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gridmm_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda"); B, heads = int(os.environ.get("ATT_B", "32")), 12
lib.gridmm_debug_att_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
names = ["prologue (Q loads issue)", "first barrier (chunk 0 + Q)", "hand-over barriers", "-", "S tiles", "softmax", "PV tiles", "epilogue"]
for (name, Sq, Sk, Wq, Wk) in [("grid self", 216, 216, 2304, 2304), ("grid x text", 216, 80, 768, 1536), ("local x kv", 57, 296, 768, 6144), ("local self", 57, 57, 2304, 2304)]:
    qb = ops.split_rows(torch.randn(B, Sq, Wq, device=dev)); kb = qb if (Wq == Wk and Sq == Sk) else ops.split_rows(torch.randn(B, Sk, Wk, device=dev))
    q = (qb.hi[..., :768], qb.lo[..., :768]); k = (kb.hi[..., Wk - 1536:Wk - 768], kb.lo[..., Wk - 1536:Wk - 768]); v = (kb.hi[..., Wk - 768:], kb.lo[..., Wk - 768:])
    mask = torch.ones(B, Sk, dtype=torch.uint8, device=dev)
    for cfg in [int(a) for a in sys.argv[1:]] or [1, 2]:
        for _ in range(2): ops.attention_rows(q, k, v, mask, cfg=cfg)
        torch.cuda.synchronize(); lib.gridmm_debug_att_prof(buf, 1)
        n = 5
        for _ in range(n): ops.attention_rows(q, k, v, mask, cfg=cfg)
        torch.cuda.synchronize(); lib.gridmm_debug_att_prof(buf, 1)
        nq, nw = {1: (1, 4), 2: (2, 4), 3: (1, 8), 5: (1, 4), 6: (2, 4)}[cfg]
        nqt = (Sq + 15) // 16
        waves = B * heads * ((nqt + nq * nw - 1) // (nq * nw)) * nw
        d = n * waves
        print("%-12s cfg %d (%d waves): cycles per wave (total %.0f): " % (name, cfg, waves, sum(buf) / d) + ", ".join("%s %.0f" % (nm, buf[i] / d) for i, nm in enumerate(names)), flush=True)
