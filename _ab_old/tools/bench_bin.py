"""grid_bin (re-bin + stable sort of the whole history) vs memory depth and slice count; checks the sliced form
against the one-workgroup kernel (perm / cell_start / cell ids identical)."""
import os, sys, numpy as np, torch
sys.path.insert(0, ".")
from gridmm_amd.grid_memory import GridMemoryBatch
from gridmm_amd import synthetic, ops
geom = synthetic.BASELINE
B = 32
def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for t in (1, 2, 3, 5, 10, 15):
    mem = GridMemoryBatch(B, geom, max_steps=t, device="cuda")
    rs = np.random.RandomState(t)
    for k in range(t):
        obs = [synthetic.make_observation(rs, geom) if hasattr(synthetic, "make_observation") else None for _ in range(B)]
        depth = np.stack([rs.randint(0, 20000, size=(geom.n_views * geom.patches ** 2)).astype(np.uint16) for _ in range(B)])
        mem.step(depth, None, [(float(rs.uniform(-3, 3)) * (k + 1) * 0.3, float(rs.uniform(-3, 3))) for _ in range(B)],
                 [float(rs.uniform(0, 6.28)) for _ in range(B)])
    ref = None
    row = []
    for S in (1, 2, 4, 8, 16):
        f = lambda: ops.grid_bin(mem.hist_x, mem.hist_y, mem.hist_valid, mem.n_pts, mem.pose_d, mem.head_d, mem.half_len,
                                 mem.cell_id, mem.perm, mem.cell_start, mem.flags, workspace=mem._bin_ws, slices=S)
        f(); torch.cuda.synchronize()
        cur = (mem.perm.clone(), mem.cell_start.clone(), mem.cell_id.clone())
        if ref is None: ref = cur
        else:
            n = int(mem.n_pts[0])
            assert torch.equal(cur[1], ref[1]) and torch.equal(cur[2], ref[2]) and torch.equal(cur[0][:, :n], ref[0][:, :n]), (t, S)
        row.append("S=%d %.1fus" % (S, t_us(f)))
    print("t=%d N=%d:" % (t, t * geom.pts_per_obs), "  ".join(row))
