"""Per-GEMM precision ablation of the navigation step: which nn.Linear of the full-size model
tolerates a 2-TERM bf16 product instead of the 3-term split  a_hi.w_hi + a_lo.w_hi + a_hi.w_lo ?

The numerics of a 2-term kernel are reproduced WITHOUT writing it: the lo plane of one operand of the chosen GEMM(s) is
zeroed before the unchanged 3-term kernel runs --
    drop a_lo:  a_hi.w_hi + a_hi.w_lo   (activations rounded to bf16, weights exact to 16 bits)
    drop w_lo:  a_hi.w_hi + a_lo.w_hi   (weights rounded to bf16)
-- for one GEMM group at a time (all other GEMMs stay 3-term), on the full-size fixture tests/golden/nav_full_b2.npz
(161 M parameters, B = 2; outputs of the imported reference).  Reported: max |logit - reference| over the four logit
sets.  GPU only:  python tools/ablate_gemm_terms.py > profiles/r2_gemm_term_ablation.txt
"""
import os, sys, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_golden
from oracle import gen_golden
from gridmm_amd import ops
import test_hip_navigation as TN

fx = load_golden("nav_full_b2.npz")
model, _ = TN._model(fx)
batch = TN._to_dev(gen_golden.full_b2_inputs())
KEYS = ("global_logits", "local_logits", "fused_logits", "grid_logits")

def err(outs):
    worst = 0.0
    for k in KEYS:
        a, w = outs[k].cpu(), torch.from_numpy(fx["out_" + k])
        f = torch.isfinite(w)
        worst = max(worst, float((a[f] - w[f]).abs().max()))
    return worst

base = err(model("navigation", batch))                     # also packs every weight: model._packed is complete now
key_of = {id(ent[1]): k for k, ent in model._packed.items()}
orig_linear = ops.linear
state = {"pat": None, "mode": None}

def patched(x, pw, *a, **kw):
    k = key_of.get(id(pw), "?")
    if state["pat"] is not None and re.fullmatch(state["pat"], k) and (pw.K % 32 == 0):
        if state["mode"] == "w_lo":
            keep = pw.lo
            pw.lo = torch.zeros_like(keep)
            try:
                return orig_linear(x, pw, *a, **kw)
            finally:
                pw.lo = keep
        act = x if isinstance(x, ops.Act) else ops.Act(x)
        if act.hi is None or not ops._is_uniform(act.hi):
            act = ops.split_rows(act.f32)
        return orig_linear(ops.Act(act.f32, act.hi, torch.zeros_like(act.lo)), pw, *a, **kw)
    return orig_linear(x, pw, *a, **kw)

ops.linear = patched
ops.TIMER = ops.KernelTimer()      # per-kernel mode: the model issues every GEMM through ops.linear (no fused-layer call)
GROUPS = [("text_proj", r"text_proj"), ("grid_proj (on the 196 reduced cells)", r"grid_proj"),
          ("grid_enc in_proj", r"grid_enc\.0\.in"), ("grid_enc out_proj", r"grid_enc\.0\.o"),
          ("grid_enc linear1 (gelu)", r"grid_enc\.0\.1"), ("grid_enc linear2", r"grid_enc\.0\.2"),
          ("grid_txt x-attn q", r"grid_txt\.0\.x\.q"), ("grid_txt x-attn kv (text)", r"grid_txt\.0\.x\.kv"),
          ("grid_txt x-attn out", r"grid_txt\.0\.x\.o"), ("grid_txt self qkv", r"grid_txt\.0\.s\.qkv"),
          ("grid_txt self out", r"grid_txt\.0\.s\.o"), ("grid_txt ffn up (gelu)", r"grid_txt\.0\.i"),
          ("grid_txt ffn down", r"grid_txt\.0\.f"), ("local K/V of all 4 layers (one GEMM)", r"local\.kv_all"),
          ("local x-attn q (4 layers)", r"local\.\d\.x\.q"), ("local x-attn out (4)", r"local\.\d\.x\.o"),
          ("local self qkv (4)", r"local\.\d\.s\.qkv"), ("local self out (4)", r"local\.\d\.s\.o"),
          ("local ffn up (4)", r"local\.\d\.i"), ("local ffn down (4)", r"local\.\d\.f"),
          ("heads (sap / fuse)", r"ghead\+lhead|gridhead|fuse|ghead|lhead"),
          ("all K/V projections", r"grid_txt\.0\.x\.kv|local\.kv_all"), ("all FFN up", r".*\.i|grid_enc\.0\.1"),
          ("all FFN down", r".*\.f|grid_enc\.0\.2"), ("EVERY GEMM", r".*")]
print("GEMM keys:", sorted(set(key_of.values())))
print("3-term everywhere: max |logit - reference| = %.2e   (test bound 2e-4; north star 1e-3)" % base)
print("%-40s %14s %14s" % ("2-term in ...", "drop a_lo", "drop w_lo"))
for name, pat in GROUPS:
    row = []
    for mode in ("a_lo", "w_lo"):
        state.update(pat=pat, mode=mode)
        row.append(err(model("navigation", batch)))
    state.update(pat=None)
    print("%-40s %14.2e %14.2e" % (name, row[0], row[1]), flush=True)
