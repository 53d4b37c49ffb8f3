"""Does a forked branch inside ONE captured graph overlap on this stack (GPU only)?  A chain of thin latency-bound GEMMs
(the local encoder's 1824-row launches) on the capture stream and one large GEMM (the K / V projection of the later
layers) on a side stream, joined at the end -- against the same launches in series.
usage: PYTHONPATH=. python tools/bench_graph_fork.py"""
import os
os.environ.setdefault("GRIDMM_LIB_DEBUG", "1")
import ctypes
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gridmm_amd import _lib, ops


def main():
    lib = _lib.load()
    dev = torch.device("cuda")

    def mk(M, N, K):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        return dict(M=M, N=N, K=K, a=ops.split_rows(x), w=ops.PackedLinear(w, b), b=b, c=torch.empty(M, N, device=dev))

    def call(p, cfg):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = lib.gridmm_linear_planes_cfg(p["a"].hi.data_ptr(), p["a"].lo.data_ptr(), p["K"], p["w"].hi.data_ptr(),
                                          p["w"].lo.data_ptr(), p["w"].Kp, p["b"].data_ptr(), None, 0, p["c"].data_ptr(), p["N"],
                                          None, None, 0, p["M"], p["N"], p["K"], 0, cfg, st)
        assert rc == 0

    thin = [mk(1824, 768, 768) for _ in range(4)] + [mk(1824, 2304, 768), mk(1824, 3072, 768), mk(1824, 768, 3072)]
    thin_cfg = [13, 13, 13, 13, 15, 15, 13]
    big = mk(6912, 4608, 768)
    front = mk(6912, 768, 768)

    def chain(cfgs=thin_cfg):
        for p, c in zip(thin, cfgs):
            call(p, c)

    def timed(fn, fork=None, reps=20):
        side = torch.cuda.Stream()
        for _ in range(2):
            fn()
            if fork:
                fork()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            cur = torch.cuda.current_stream()
            for _ in range(reps):
                call(front, 15)              # a common predecessor
                if fork:
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        fork()
                fn()
                if fork:
                    cur.wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
        return best

    t_front = timed(lambda: None)
    t_chain = timed(chain)
    print("front alone %.1f us; front + thin chain (7 launches) %.1f us" % (t_front, t_chain))
    for bcfg in (36, 16, 15):
        t_big = timed(lambda: call(big, bcfg))
        t_ser = timed(lambda: (call(big, bcfg), chain()))
        t_fork = timed(chain, fork=lambda: call(big, bcfg))
        print("big cfg %d: front + big %.1f | serial front + big + chain %.1f | forked %.1f  (gain %.1f us)" %
              (bcfg, t_big, t_ser, t_fork, t_ser - t_fork))
    # thin chain with small-LDS tiles (64 KB: co-resident with a 96-KB 256x128 workgroup)
    small = [43, 43, 43, 43, 4, 4, 43]
    t_ser = timed(lambda: (call(big, 16), chain(small)))
    t_fork = timed(lambda: chain(small), fork=lambda: call(big, 16))
    print("big cfg 16 + 64x64 thin tiles: serial %.1f | forked %.1f" % (t_ser, t_fork))
    # two thin chains side by side (episode halves would look like this)
    thin2 = [mk(1824, 768, 768) for _ in range(4)] + [mk(1824, 2304, 768), mk(1824, 3072, 768), mk(1824, 768, 3072)]

    def chain2():
        for p, c in zip(thin2, thin_cfg):
            call(p, c)
    t_ser = timed(lambda: (chain(), chain2()))
    t_fork = timed(chain, fork=chain2)
    print("two independent thin chains: serial %.1f | forked %.1f" % (t_ser, t_fork))
    # an empty fork (cost of the fork / join itself)
    tiny = mk(128, 64, 64)
    t_ser = timed(lambda: (call(tiny, 4), chain()))
    t_fork = timed(chain, fork=lambda: call(tiny, 4))
    print("fork / join cost with a tiny side kernel: serial %.1f | forked %.1f" % (t_ser, t_fork))


if __name__ == "__main__":
    main()
