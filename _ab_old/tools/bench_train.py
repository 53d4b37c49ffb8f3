"""Timing of one pre-training step (forward + backward + clip + AdamW) at the full model size on one GPU.
Secondary measurement (the headline metric is the inference nav step, bench.py); SURVEY.md §8 config 3 shape:
B=32 per GPU, native grid (588*t points x 768)."""
import argparse
import json
import time

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--task", default="sap")
    a = ap.parse_args()
    from gridmm_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    from gridmm_amd.pretrain_loop import PreTrainer, default_opts
    from gridmm_amd.synthetic import batch_to, make_pretrain_batch
    from gridmm_amd.vilmodel import default_config
    from gridmm_amd import ops
    cfg = default_config(use_lang2visn_attn=True, pretrain_tasks=["mlm", "mrc", "sap"], image_prob_size=1000,
                         obj_prob_size=0)
    torch.manual_seed(0)
    model = GlocalTextPathCMTPreTraining(cfg).cuda()
    tr = PreTrainer(model, default_opts(warmup_steps=100))
    batch = batch_to(make_pretrain_batch(np.random.RandomState(0), a.batch, a.task, max_steps=5, L=80, vocab=30000,
                                         image_prob_size=1000, n_pts=(588 * 3, 588 * 5)), "cuda")
    for _ in range(a.warmup):
        tr.train_step(batch, a.task)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.train_step(batch, a.task)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    # kernel-class breakdown of one more step
    ops.TIMER = ops.KernelTimer()
    tr.train_step(batch, a.task)
    torch.cuda.synchronize()
    summ = {k: round(v["ms"], 3) for k, v in ops.TIMER.summary().items()}
    ops.TIMER = None
    print(json.dumps({"task": a.task, "batch": a.batch, "ms_per_step": round(dt * 1e3, 2),
                      "samples_per_s": round(a.batch / dt, 1), "timed_kernel_ms": summ}))


if __name__ == "__main__":
    main()
