"""In-kernel cycle stamps of grid_relevance_gemm_kernel (build aggregate_relg.hip with -DGRIDMM_RELG_PROF)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gridmm_amd import ops, _lib
from gridmm_amd.grid_memory import pack_reference_lists
B, n, D = 32, 7056, 512
L = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(0)
fts = [(torch.randn(n, D, device="cuda") * 0.5).half() for _ in range(B)]
maps = [torch.from_numpy(rng.integers(0, 196, size=n)).double().cuda() for _ in range(B)]
slab, perm, cs = pack_reference_lists(fts, maps)
frag = ops.text_fragments(torch.randn(B, L, D, device="cuda") * 0.3)
for _ in range(3):
    ops.grid_aggregate(slab, perm, cs, frag, L)
torch.cuda.synchronize()
lib = _lib.load()
out = (ctypes.c_longlong * 64)()
lib.gridmm_debug_relg_prof.argtypes = [ctypes.c_void_p]
assert lib.gridmm_debug_relg_prof(out) == 0
a = np.array(out).reshape(8, 8)
print("L=%d  per wave: total | wait | barrier | issue | mfma  (cycles per k-step), steps" % L)
for w in range(4):
    st = max(1, a[w, 5])
    print(w, a[w, 0], " ".join("%7.0f" % (a[w, k] / st) for k in (0, 1, 2, 3, 4)), a[w, 5])
