"""Tile-configuration sweep inside the CAPTURED pre-training step (one task): records the GEMM shapes the step contains
(gridmm_debug_gemm_shapes), then for the shapes that carry the most launches forces each candidate configuration,
re-captures the step and times its replays, alternating with the heuristic (A B A B).
usage: PYTHONPATH=. python tools/sweep_gemm_cfg_train.py [task] [n_shapes]"""
import os
os.environ.setdefault("GRIDMM_LIB_DEBUG", "1")   # development build: tile overrides + the whole experiment table (make -C gridmm_amd/csrc debug)
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gridmm_amd import _lib

CANDS = [43, 13, 8, 15, 2, 14, 16, 36, 50]


def main():
    task = sys.argv[1] if len(sys.argv) > 1 else "sap"
    n_shapes = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    from gridmm_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    from gridmm_amd.pretrain_loop import PreTrainer, default_opts
    from gridmm_amd.synthetic import batch_to, make_pretrain_batch
    from gridmm_amd.train_graph import GraphedTrainStep
    from gridmm_amd.vilmodel import default_config
    lib = _lib.load()
    dev = torch.device("cuda")
    cfg = default_config(use_lang2visn_attn=True, pretrain_tasks=["mlm", "mrc", "sap"], image_prob_size=1000, obj_prob_size=0)
    torch.manual_seed(0)
    model = GlocalTextPathCMTPreTraining(cfg).to(dev)
    tr = PreTrainer(model, default_opts(warmup_steps=100))
    batch = batch_to(make_pretrain_batch(np.random.RandomState(1), 32, task, max_steps=5, L=80, vocab=30000, image_prob_size=1000,
                                         n_pts=(588 * 3, 588 * 5)), dev)
    for _ in range(2):
        tr.train_step(batch, task)
    lib.gridmm_debug_gemm_shapes(None, 0, 1)
    tr.train_step(batch, task)
    buf = (ctypes.c_int * (128 * 5))()
    n = lib.gridmm_debug_gemm_shapes(buf, 128, 0)
    shapes = sorted([tuple(buf[i * 5:i * 5 + 5]) for i in range(n)], key=lambda r: -r[4] * r[0] * r[1] * r[2])
    print("%d GEMM shapes in one eager step (M, N, K, cfg, calls), by flops:" % n)
    for r in shapes:
        print("  ", r)

    def timed(reps=12):
        g = GraphedTrainStep(tr, batch, task)
        for _ in range(2):
            g()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g()
        e1.record()
        torch.cuda.synchronize()
        del g
        return e0.elapsed_time(e1) * 1e3 / reps

    base = timed()
    print("heuristic: %.1f us per captured step (%s)" % (base, task), flush=True)
    for (M, N, K, cfg0, calls) in shapes[:n_shapes]:
        line = "%5d x %4d x %4d  cfg %2d x%3d |" % (M, N, K, cfg0, calls)
        for c in CANDS:
            if c == cfg0:
                continue
            try:
                lib.gridmm_debug_gemm_cfg_override(M, N, K, 0)
                tb = timed()
                lib.gridmm_debug_gemm_cfg_override(M, N, K, c)
                tc = timed()
                line += " %d:%+.0f" % (c, tc - tb)
            except Exception as e:
                line += " %d:x" % c
                torch.cuda.synchronize()
            lib.gridmm_debug_gemm_cfg_override(M, N, K, 0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
