"""Diagnostic: is the pre-training step bit-reproducible?  Runs the same two tasks twice (eager / captured, per argv[1]) from
identical copies of the model and prints whether losses and gradient norms match bit for bit (the property
tests/test_hip_train_graph.py asserts; this script is the interactive form used to find the atomics that broke it in round 3).
usage (repo root): python tools/dbg_determinism.py eager|graph"""
import os, sys, copy
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from train_graph_cases import _setup
from gridmm_amd.pretrain_loop import PreTrainer, default_opts
from gridmm_amd.train_graph import GraphedTrainStep
mode = sys.argv[1]
model, batches = _setup(0.0)
if os.environ.get('FP32_GRID_PROJ'):
    model.bert.grid_proj.float()
tasks = ("mlm", "sap")
def run_eager():
    m = copy.deepcopy(model); tr = PreTrainer(m, default_opts(warmup_steps=10)); out = []
    for t in tasks:
        for _ in range(2): tr.train_step(batches[t], t)
    for i in range(8):
        t = tasks[i % 2]; l, n = tr.train_step(batches[t], t); out.append(l.clone())
    return out
def run_graph():
    m = copy.deepcopy(model); tr = PreTrainer(m, default_opts(warmup_steps=10)); out = []
    gs = {t: GraphedTrainStep(tr, batches[t], t) for t in tasks}
    for i in range(8):
        t = tasks[i % 2]; l, n = gs[t](); out.append(l.clone())
    return out
f = run_eager if mode == "eager" else run_graph
ref = f()
for rep in range(4):
    cur = f()
    print(mode, rep, ["%.1e" % float((a - b).abs().max()) for a, b in zip(ref, cur)])
