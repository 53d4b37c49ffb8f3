"""Split-K probe for the small-M FFN-down GEMM (M x 768 x 3072): plain launch vs gridmm_linear_planes_splitk."""
import sys, torch
sys.path.insert(0, ".")
from gridmm_amd import ops, _lib
from gridmm_amd.ops import _p, _stream
lib = _lib.load()
def t_us(fn, n=40):     # device time: n calls captured in one hipGraph
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for M, N, K in ((1824, 768, 3072), (1824, 768, 768), (6912, 768, 3072), (6912, 768, 768)):
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.02
    a = ops.split_rows(x)
    pw = ops.PackedLinear(w, None)
    c0 = torch.empty(M, N, device="cuda")
    base = t_us(lambda: ops.linear(a, pw, out=c0))
    ref = ops.linear(a, pw).f32
    res = [("plain", base)]
    wh, wl = pw.hi, pw.lo
    for splits in (2, 3, 4, 6, 8, 12):
        if (K // 32) % splits: continue
        out = torch.empty(M, N, device="cuda"); ws = torch.empty(splits, M, N, device="cuda")
        f = lambda: _lib.check(lib.gridmm_linear_planes_splitk(_p(a.hi), _p(a.lo), K, _p(wh), _p(wl), pw.Kp, _p(out), _p(ws), M, N, K, splits, _stream()), "sk")
        f(); torch.cuda.synchronize()
        err = float((out - ref).abs().max())
        res.append(("x%d" % splits, t_us(f), err))
    print(M, N, K, [(r[0], round(r[1], 1)) + tuple(r[2:]) for r in res])
