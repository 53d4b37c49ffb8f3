"""Tile-configuration sweep INSIDE the captured B = 32 step: for a GEMM shape of the step (or the attention launch shape),
force each candidate configuration (gridmm_debug_gemm_cfg_override / gridmm_debug_attention_cfg_override), re-capture the
step and time its replays, alternating with the heuristic's choice (A B A B: boxes drift by ~10 us within a minute).
The isolated micro-benchmark (tools/bench_gemm.py) sees operands that differ from the step's: here A was just written by
the previous launch and the L2s were flushed at the kernel boundary.
usage: PYTHONPATH=. python tools/sweep_gemm_cfg_step.py [thin | all | attention] ..."""
import os
os.environ.setdefault("GRIDMM_LIB_DEBUG", "1")   # development build: tile overrides + the whole experiment table (make -C gridmm_amd/csrc debug)
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gridmm_amd import _lib

SHAPES = {  # (M, N, K): (what, launches per step)
    (2560, 512, 768): ("text_proj", 1), (6272, 768, 512): ("grid_proj", 1), (6912, 2304, 768): ("QKV grid", 2),
    (6912, 768, 768): ("out / q proj grid", 5), (6912, 3072, 768): ("FFN1 grid", 2), (6912, 768, 3072): ("FFN2 grid", 2),
    (2560, 1536, 768): ("text K/V", 1), (9472, 6144, 768): ("local K/V", 1), (1824, 768, 768): ("local q / out", 12),
    (1824, 2304, 768): ("local QKV", 4), (1824, 3072, 768): ("local FFN1", 4), (1824, 768, 3072): ("local FFN2", 4)}
THIN = [(1824, 768, 768), (1824, 768, 3072), (1824, 2304, 768), (1824, 3072, 768), (2560, 512, 768), (2560, 1536, 768)]
CANDS_ALL = [43, 8, 4, 6, 9, 13, 21, 15, 14, 1, 2, 12, 16, 3, 36, 42, 44]
BIG = [(6912, 3072, 768), (6912, 768, 3072), (6912, 768, 768), (6912, 2304, 768), (9472, 6144, 768), (1824, 2304, 768),
       (1824, 3072, 768), (6272, 768, 512)]
CANDS_BIG = [15, 36, 60, 61, 62, 63, 64, 16]        # BK = 32 tiles, all with tiled weight planes
CANDS_THIN = [43, 8, 13, 21, 50, 52, 53, 55, 56, 6, 9, 15]
MID = [(6912, 768, 768), (6912, 768, 3072), (6912, 3072, 768), (6272, 768, 512), (2560, 1536, 768)]
CANDS_MID = [65, 66, 70]                        # round 6: one fat workgroup per CU, deeper rings (80-120 KB in flight)
ATT_BIG = [9, 3, 20, 21, 22, 23, 24, 25]
ATT_SMALL = [5, 1, 15, 18]


def main():
    modes = sys.argv[1:] or ["thin"]
    lib = _lib.load()
    dev = torch.device("cuda")
    args = argparse.Namespace(batch=32, shape="baseline", mem_steps=1, eager=False)
    model, batch, mem, eps, step, eager_step, geom = bench.build_workload(args, dev)
    gs = step.graph

    def timed(n=30):
        for _ in range(2):
            gs._device_step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            outs = gs._device_step()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ref = outs["fused_logits"].clone()
        del g
        return e0.elapsed_time(e1) * 1e3 / n, ref

    base, ref0 = timed()
    print("heuristic: %.1f us per step" % base, flush=True)

    def ab(set_cand, set_base, reps=3):
        """Mean of t(candidate) - t(heuristic) over `reps` alternations."""
        d, same = [], True
        for _ in range(reps):
            set_base()
            tb, _ = timed(20)
            set_cand()
            tc, ref = timed(20)
            same = same and torch.equal(ref, ref0)
            d.append(tc - tb)
        set_base()
        return sum(d) / len(d), same

    winners = {}
    for mode in modes:
        if mode == "agg":        # workgroups per episode of the aggregation launch (ops.grid_aggregate reads the variable per call)
            line = "aggregation chunks per episode (default 256 / B = 8) |"
            for c in (4, 6, 8, 10, 12, 16, 24):
                def on(c=c):
                    os.environ["GRIDMM_AGG_CHUNKS"] = str(c)
                def off():
                    os.environ.pop("GRIDMM_AGG_CHUNKS", None)
                d, same = ab(on, off)
                line += " %d:%+.1f%s" % (c, d, "" if same else "(!)")
            print(line, flush=True)
            continue
        if mode == "attention":
            for name, cands, setter in (("> 4 query tiles", ATT_BIG, lambda c: lib.gridmm_debug_attention_cfg_override(c, 0)),
                                        ("<= 4 query tiles", ATT_SMALL, lambda c: lib.gridmm_debug_attention_cfg_override(0, c))):
                line = "attention, %-18s |" % name
                for c in cands:
                    try:
                        d, same = ab(lambda: setter(c), lambda: setter(0))
                        line += " %d:%+.1f%s" % (c, d, "" if same else "(!)")
                    except Exception as e:
                        line += " %d:x" % c
                        torch.cuda.synchronize()
                print(line, flush=True)
            continue
        shapes = THIN if mode == "thin" else (BIG if mode == "big" else (MID if mode == "mid" else list(SHAPES)))
        cands = CANDS_THIN if mode == "thin" else (CANDS_BIG if mode == "big" else (CANDS_MID if mode == "mid" else CANDS_ALL))
        if mode == "ffn1":
            shapes, cands = [(6912, 3072, 768), (6912, 2304, 768), (9472, 6144, 768)], [76, 15, 36]
        if mode == "ffn1x":
            shapes, cands = [(6912, 3072, 768), (1824, 3072, 768), (1824, 2304, 768)], [76, 76, 76, 15]
        if mode == "thin6":
            shapes, cands = [(1824, 768, 768), (1824, 768, 3072), (2560, 512, 768)], [71, 75]
        for (M, N, K) in shapes:
            what, cnt = SHAPES[(M, N, K)]
            line = "%5d x %4d x %4d  %-18s x%2d |" % (M, N, K, what, cnt)
            res = {}
            for cfg in cands:
                try:
                    d, same = ab(lambda: lib.gridmm_debug_gemm_cfg_override(M, N, K, cfg),
                                 lambda: lib.gridmm_debug_gemm_cfg_override(M, N, K, 0), reps=2)
                    res[cfg] = d
                    line += " %d:%+.1f%s" % (cfg, d, "" if same else "(!)")
                except Exception as e:           # a configuration this shape cannot take (K % 64, ...)
                    line += " %d:x" % cfg
                    torch.cuda.synchronize()
                    lib.gridmm_debug_gemm_cfg_override(M, N, K, 0)
            if res:
                c = min(res, key=res.get)
                winners[(M, N, K)] = (c, res[c])
            print(line, flush=True)
    if winners:
        print("best per shape (us per step vs the heuristic, A/B alternated):")
        for k, (c, d) in winners.items():
            print("  %s -> cfg %d  %+.1f us" % (k, c, d))

        def all_on():
            for k, (c, d) in winners.items():
                if d < -2.0:
                    lib.gridmm_debug_gemm_cfg_override(k[0], k[1], k[2], c)

        def all_off():
            for k in winners:
                lib.gridmm_debug_gemm_cfg_override(k[0], k[1], k[2], 0)
        d, same = ab(all_on, all_off, reps=3)
        print("all winners (< -2 us) together: %+.1f us per step vs heuristic %.1f  logits equal: %s" % (d, base, same))


if __name__ == "__main__":
    main()
