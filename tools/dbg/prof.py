import os, sys
sys.path.insert(0, os.getcwd())
os.environ["GRIDMM_AGG_PROF"] = "1"
import numpy as np, torch
from gridmm_amd import ops, _lib
B, N, D, L = 32, 7056, 512, 80
rs = np.random.RandomState(0)
slab = torch.from_numpy((rs.standard_normal((B, N, D)) * 0.35).astype(np.float16)).cuda()
ids = torch.from_numpy(rs.randint(0, 196, size=(B, N)).astype(np.int16)).cuda()
perm = torch.empty(B, N, dtype=torch.int32, device="cuda"); cs = torch.empty(B, 198, dtype=torch.int32, device="cuda")
ops.grid_sort_ids(ids, torch.full((B,), N, dtype=torch.int32, device="cuda"), perm, cs)
text = torch.randn(B, L, D, device="cuda") * 0.3
frag = ops.text_fragments(text)
lib = _lib.load()
import ctypes
n_chunks = 8
cells = torch.zeros(B * 196 * D + 64, dtype=torch.float32, device="cuda")
occ = torch.empty(B, 196, dtype=torch.uint8, device="cuda")
chunks = torch.empty(B, n_chunks + 1, dtype=torch.int32, device="cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
for _ in range(3):
    st = lib.gridmm_grid_aggregate(p(slab), p(perm), p(cs), p(frag), p(cells), p(occ), None, p(chunks), B, N, D, L, n_chunks, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
torch.cuda.synchronize()
o = cells[B * 196 * D:B * 196 * D + 32].cpu().numpy().reshape(8, 4)
print("wave: wait  issue  work  (cycles per tile), ntiles")
for w in range(8):
    n = o[w, 3] + 1
    print(w, (o[w, :3] / n).round(0), int(o[w, 3]))
