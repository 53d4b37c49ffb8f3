"""Split-K probe for the small-M FFN-down GEMM (M x 768 x 3072): plain launch vs gridmm_linear_planes_splitk."""
import sys, torch
sys.path.insert(0, ".")
from gridmm_amd import ops, _lib
from gridmm_amd.ops import _p, _stream
lib = _lib.load()
def t_us(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for M, N, K in ((1824, 768, 3072), (1824, 768, 768), (6912, 768, 3072)):
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.02
    a = ops.split_rows(x)
    pw = ops.PackedLinear(w, None)
    base = t_us(lambda: ops.linear(a, pw))
    ref = ops.linear(a, pw).f32
    res = [("plain", base)]
    wh, wl = pw.hi, pw.lo
    for splits in (2, 3, 4, 6):
        if (K // 32) % splits: continue
        out = torch.empty(M, N, device="cuda"); ws = torch.empty(splits, M, N, device="cuda")
        f = lambda: _lib.check(lib.gridmm_linear_planes_splitk(_p(a.hi), _p(a.lo), K, _p(wh), _p(wl), pw.Kp, _p(out), _p(ws), M, N, K, splits, _stream()), "sk")
        f(); torch.cuda.synchronize()
        err = float((out - ref).abs().max())
        res.append(("x%d" % splits, t_us(f), err))
    print(M, N, K, res)
