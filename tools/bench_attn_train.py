"""Micro-benchmark (device time, hipGraph of 20 calls): forward and backward of the training attention at the shapes of the
pre-training / fine-tune step -- exact-fp32 kernels (gridmm_attention_train_planes / gridmm_attention_bwd) vs the bf16
matrix-pipe kernels of round 5 (gridmm_attention_rows_train / gridmm_attention_rows_bwd).  B x 12 heads, dropout 0.1."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gridmm_amd import _lib, ops

SHAPES = [(32, 80, 80), (32, 216, 216), (32, 57, 296), (32, 57, 57), (32, 37, 37), (32, 216, 80), (32, 300, 300)]


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def graph_time(fn, n=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def main():
    lib = _lib.load()
    dev = torch.device("cuda")
    heads, H, p, seed = 12, 768, 0.1, 12345
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)   # noqa: E731
    for (B, Sq, Sk) in SHAPES:
        torch.manual_seed(0)
        q = torch.randn(B, Sq, H, device=dev)
        kv = torch.randn(B, Sk, 2 * H, device=dev)
        dy = torch.randn(B, Sq, H, device=dev)
        km = torch.ones(B, Sk, dtype=torch.uint8, device=dev)
        qa, ka = ops.split_rows(q), ops.split_rows(kv)
        Sqp = (Sq + 15) // 16 * 16
        out = torch.empty(B, Sq, H, device=dev)
        oh, ol = ops._planes_like(out.shape, dev)
        lse = torch.empty(B, heads, Sqp, device=dev)
        delta = torch.empty_like(lse)
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        ws = torch.empty(lib.gridmm_attention_rows_bwd_workspace(B, heads, Sq), dtype=torch.uint8, device=dev)
        k32, v32 = kv[..., :H], kv[..., H:]
        off = lambda t, c: ctypes.c_void_p(t.data_ptr() + 2 * c)    # noqa: E731
        foff = lambda t, c: ctypes.c_void_p(t.data_ptr() + 4 * c)   # noqa: E731

        def f32():
            assert lib.gridmm_attention_train_planes(_p(q), Sq * H, H, _p(k32), Sk * 2 * H, 2 * H, foff(kv, H), Sk * 2 * H, 2 * H, _p(km), Sk,
                                                     _p(out), Sq * H, H, _p(oh), _p(ol), Sq * H, H, _p(lse), Sqp, B, heads, Sq, Sk, 0.125, p,
                                                     seed, None, st()) == 0

        def b32():
            assert lib.gridmm_attention_bwd(_p(q), Sq * H, H, _p(k32), Sk * 2 * H, 2 * H, foff(kv, H), Sk * 2 * H, 2 * H, _p(km), Sk, _p(out),
                                            Sq * H, H, _p(dy), Sq * H, H, _p(lse), _p(delta), _p(dq), Sq * H, H, _p(dkv), Sk * 2 * H, 2 * H,
                                            foff(dkv, H), Sk * 2 * H, 2 * H, B, heads, Sq, Sk, Sqp, 0.125, p, seed, None, st()) == 0

        def f16():
            assert lib.gridmm_attention_rows_train(_p(qa.hi), _p(qa.lo), Sq * H, H, _p(ka.hi), _p(ka.lo), Sk * 2 * H, 2 * H, off(ka.hi, H),
                                                   off(ka.lo, H), Sk * 2 * H, 2 * H, _p(km), Sk, _p(out), Sq * H, H, _p(oh), _p(ol), Sq * H, H,
                                                   _p(lse), Sqp, B, heads, Sq, Sk, 0.125, p, seed, None, st()) == 0

        def b16():
            assert lib.gridmm_attention_rows_bwd(_p(qa.hi), _p(qa.lo), Sq * H, H, _p(ka.hi), _p(ka.lo), Sk * 2 * H, 2 * H, off(ka.hi, H),
                                                 off(ka.lo, H), Sk * 2 * H, 2 * H, _p(km), Sk, _p(out), Sq * H, H, _p(dy), Sq * H, H, _p(lse),
                                                 _p(ws), ws.numel(), _p(dq), Sq * H, H, _p(dkv), Sk * 2 * H, 2 * H, foff(dkv, H), Sk * 2 * H,
                                                 2 * H, B, heads, Sq, Sk, Sqp, 0.125, p, seed, None, st()) == 0
        t = [graph_time(f) for f in (f32, b32)]
        f16()
        t += [graph_time(f) for f in (f16, b16)]
        print("B=%d Sq=%3d Sk=%3d | fp32 fwd %6.1f us bwd %6.1f us | bf16x3 fwd %6.1f us bwd %6.1f us" % (B, Sq, Sk, *t), flush=True)


if __name__ == "__main__":
    main()
