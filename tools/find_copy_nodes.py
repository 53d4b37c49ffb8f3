"""Which torch operations of the (taped) pre-training step become memcpy / memset NODES when the step is captured?
The library's own launches are all kernels; the runtime's pre-recorded graph packets mishandle the few copy / fill nodes
torch adds ("training-graph fault").  Runs the taped step eagerly under torch.profiler and prints every
device memcpy / memset with the python frames that issued it.     usage: python tools/find_copy_nodes.py [task ...]"""
import os
import sys

sys.path.insert(0, ".")
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

from gridmm_amd import hostsync as hs
from gridmm_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
from gridmm_amd.pretrain_loop import PreTrainer, default_opts
from gridmm_amd.synthetic import batch_to, make_pretrain_batch
from gridmm_amd.vilmodel import default_config

dev = torch.device("cuda:0")
tasks = sys.argv[1:] or ["mlm", "mrc", "sap"]
cfg = default_config(use_lang2visn_attn=True, pretrain_tasks=["mlm", "mrc", "sap"], image_prob_size=1000, obj_prob_size=0)
torch.manual_seed(0)
model = GlocalTextPathCMTPreTraining(cfg).to(dev)
tr = PreTrainer(model, default_opts(warmup_steps=100))
for i, t in enumerate(tasks):
    batch = batch_to(make_pretrain_batch(np.random.RandomState(i), 32, t, max_steps=5, L=80, vocab=30000, image_prob_size=1000,
                                         n_pts=(588 * 3, 588 * 5)), dev)
    with hs.record() as tape:
        tr.train_step(batch, t)
    with hs.replay(tape):
        tr.train_step(batch, t)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        with hs.replay(tape):
            model.train()
            losses = model(batch, task=t, compute_loss=True)
            losses.mean().backward()
        torch.cuda.synchronize()
    tr.optimizer.zero_grad(set_to_none=True)
    path = "/tmp/find_copy_nodes_%s.json" % t
    prof.export_chrome_trace(path)
    import json
    tr_events = json.load(open(path))["traceEvents"]
    gpu = [e for e in tr_events if e.get("cat") in ("gpu_memcpy", "gpu_memset")]
    rt = {e["args"].get("correlation"): e for e in tr_events if e.get("cat") in ("cuda_runtime", "cuda_driver") and "args" in e}
    ops = [e for e in tr_events if e.get("cat") == "cpu_op"]
    pyf = [e for e in tr_events if e.get("cat") == "python_function"]
    print("== task %s: %d device memcpy / memset activities" % (t, len(gpu)), flush=True)
    for g in gpu:
        r = rt.get(g["args"].get("correlation"))
        line = "  %-26s %7.1f us bytes=%s" % (g["name"][:26], g.get("dur", 0), g["args"].get("bytes"))
        if r is None:
            print(line + "  (no runtime record)")
            continue
        ts, tid = r["ts"], r["tid"]
        encl = sorted([o for o in ops if o["tid"] == tid and o["ts"] <= ts <= o["ts"] + o.get("dur", 0)], key=lambda o: o["ts"])
        print(line + "  api=%s ops=%s" % (r["name"], [o["name"] for o in encl][-4:]))
        fr = sorted([f for f in pyf if f["tid"] == tid and f["ts"] <= ts <= f["ts"] + f.get("dur", 0)], key=lambda f: f["ts"])
        for f in [f for f in fr if "gridmm_amd" in f["name"] or "tools/" in f["name"]][-4:]:
            print("        " + f["name"][:160])
