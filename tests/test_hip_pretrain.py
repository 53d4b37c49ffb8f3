"""GPU: the pre-training twin (gridmm_amd.pretrain_cmt) against vectors captured from the imported reference
(pretrain_src/model/pretrain_cmt.py; oracle/gen_golden.py gen_pretrain): per-sample loss vectors of mlm / mrc / sap,
and -- after loss.mean().backward() as train_r2r.py:245-262 does -- every parameter's gradient norm, 48 seeded
samples of every gradient, and the set of parameters without a gradient.

Tolerances (measured: tools/grad_errs.py; median gradient error 1e-5 .. 7e-5): losses agree to 2e-3 relative (measured
<= 1.4e-5); every gradient to 5e-3 of the tensor's largest sampled entry / norm, EXCEPT text_proj.{weight,bias} (2e-2;
measured 0.4-1.5 %): the reference takes the instruction-relevance product, grid_proj and the per-cell reduction in fp16
(pretrain_src/model/vilmodel.py:664,690-703), this build in fp32 / f16 hi+lo -- the relevance values differ at the fp16
rounding level, which moves a few arg-max routes of the max over tokens, and text_proj is the only parameter whose whole
gradient flows through that routing.

mrc: RegionClassification holds a ReLU (pretrain_cmt.py:15-18).  In rounds 1-3 one of the 7680 pre-activations of the
reduced fixture sat within 1e-4 of zero (perturbing the REFERENCE's own weights by 1e-4 relative moved its mrc gradients by
2-8 %), and mrc was checked at 1e-1 + cosine.  Round 4 regenerated the reduced mrc fixtures from batch seeds whose
pre-activations all clear the gate by >= 1.05e-3 / 6.6e-4 (oracle/search_pretrain_seeds.py, gen_golden.PRETRAIN_SEEDS): no
gate can flip, and mrc is pinned elementwise at 5e-3 like every other task (the cosine check stays).

text_proj, round 6 -- two causes, separated.  (i) Arg-max near-ties: the fixtures carry the number of grid points whose two
best instruction tokens are closer than 4x the difference between the reference's half x half relevance product and the same
product with un-rounded text features (`relevance_ties_<task>` = [near-tie points, points], oracle/search_relevance_ties.py):
14-29 of 1 000-1 500 points in the reduced fixtures, HALF of the points at the released size (9-layer text encoder: the token
features are nearly parallel).  pretrain_reduced_notie.npz (40-64 points per episode, batch seeds searched for ZERO near-ties,
min top-2 gap 5.8 .. 12.4x the discrepancy; sap / mrc seeds also clear every ReLU gate) removes that cause: text_proj.weight
comes out at 5.4e-3 (mlm), < 5e-3 (mrc) -- and still 1.1e-2 for sap.  (ii) The reference's backward through the grid path runs
in fp16 (grid_proj output, softmax weights and per-cell sums are half tensors, pretrain_src/model/vilmodel.py:690-703), and
d(loss)/d(grid_proj output) lies largely in fp16's SUBNORMAL range: `fp16_grad_<task>` = [max, median of the non-zero entries,
share exactly zero, share below 6.1e-5] -- 15-60 % of the non-zero entries are subnormal (a few significant bits), median 7e-6
.. 4e-4.  text_proj's gradient is a sum of exactly those quantised values; this build carries them in fp32.  So the bound for
text_proj stays 2e-2 wherever near-ties exist or the task is sap, 7e-3 for the tie-free mlm / mrc batches; every other
parameter 5e-3 everywhere.
"""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import gen_golden
from oracle.ref_harness import det_tensor

pytestmark = pytest.mark.gpu


def _model(fx):
    from gridmm_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    from gridmm_amd.vilmodel import default_config
    cfg = default_config(**json.loads(str(fx["cfg"])))
    m = GlocalTextPathCMTPreTraining(cfg)
    dt = json.loads(str(fx["param_dtypes"]))
    sd = {}
    for k, v in m.state_dict().items():
        assert k in dt, k
        sd[k] = det_tensor(k, v.shape, int(fx["weight_seed"])).to(v.dtype)
    assert sorted(sd) == sorted(dt), set(sd) ^ set(dt)                       # same state_dict keys as the reference
    for k in sd:
        assert str(sd[k].dtype) == dt[k], (k, sd[k].dtype, dt[k])
    if "mlm_head.predictions.decoder.weight" in sd:
        sd["mlm_head.predictions.decoder.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    m.load_state_dict(sd)
    assert [k for k, _ in m.named_parameters()] == json.loads(str(fx["param_names"]))
    return m.cuda().train()          # dropout probabilities are 0 in the reduced config


@pytest.mark.parametrize("task,with_obj", [("mlm", False), ("mrc", False), ("sap", False),
                                           ("mrc", True), ("sap", True), ("og", True),
                                           ("mlm", "full"), ("mrc", "full"), ("sap", "full"),
                                           ("mlm", "notie"), ("mrc", "notie")])
def test_pretrain_losses_and_gradients_match_reference(task, with_obj):
    """with_obj == "full": the released full-size configuration (9 / 2 / 4 layers, 3072-wide FFN, 30 522-word vocabulary,
    161 M parameters), B = 2 -- tests/golden/pretrain_full_b2.npz from the imported reference.
    with_obj == "notie": pretrain_reduced_notie.npz -- no arg-max near-tie in the relevance product: text_proj at 7e-3 (mlm,
    mrc).  The fixture also holds the sap task; it is not asserted: without any near-tie its text_proj gradient is 1.1e-2 from
    the reference (the fp16 backward, module docstring) and the ReLU heads of sap leave one LayerNorm bias at 7.5e-3."""
    from gridmm_amd.synthetic import batch_to
    full = with_obj == "full"
    notie = with_obj == "notie"
    with_obj = with_obj is True
    fx = load_golden("pretrain_reduced_notie.npz" if notie else
                     ("pretrain_full_b2.npz" if full else ("pretrain_reduced_obj.npz" if with_obj else "pretrain_reduced.npz")))
    ties = fx["relevance_ties_" + task] if ("relevance_ties_" + task) in fx.files else None
    if notie:
        assert ties is not None and int(ties[0]) == 0 and int(ties[1]) >= 120
        gap, disc = fx["relevance_gap_" + task]
        assert gap >= 4.0 * disc          # (the search's criterion, per episode; 5.8 .. 12.4 on the three batches)
    if ("fp16_grad_" + task) in fx.files:     # the reference's own precision on this path (see the module docstring)
        assert float(fx["fp16_grad_" + task][3]) > 0.10, "share of fp16-subnormal gradient entries in the reference's grid path"
    elif ties is not None:
        assert int(ties[0]) >= 10, "the tie-prone fixture is expected to hold near-ties (that is why its bound is 2e-2)"
        print("near-tie points: %d of %d" % (int(ties[0]), int(ties[1])))
    model = _model(fx)
    batch = batch_to(gen_golden.pretrain_notie_batch(task) if notie else
                     (gen_golden.pretrain_full_batch(task) if full else gen_golden.pretrain_batch(task, with_obj)), "cuda")
    loss = model(batch, task=task, compute_loss=True)
    want = fx["loss_" + task]
    got = loss.detach().cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 2e-3 * max(1.0, np.abs(want).max()), (got, want)

    loss.mean().backward()
    names = json.loads(str(fx["grad_names_" + task]))
    params = dict(model.named_parameters())
    with_grad = [k for k, p in params.items() if p.grad is not None]
    assert sorted(with_grad) == sorted(names), set(with_grad) ^ set(names)     # same grad-less parameters
    norms, samples = fx["grad_norms_" + task], fx["grad_samples_" + task]
    scale = float(norms.max())
    o, errs, got_all, ref_all = 0, [], [], []
    for k, n_ref in zip(names, norms):
        g = params[k].grad.detach().float().reshape(-1).cpu()
        idx = gen_golden.grad_sample_index(k, g.numel())
        ref = samples[o:o + len(idx)]
        o += len(idx)
        got_all.append(g[torch.from_numpy(idx)].numpy())
        ref_all.append(ref)
        denom = max(float(np.abs(ref).max()), 1e-3 * scale / np.sqrt(max(g.numel(), 1)), 1e-12)
        e = float(np.abs(g[torch.from_numpy(idx)].numpy() - ref).max()) / denom
        en = abs(float(g.norm()) - float(n_ref)) / max(float(n_ref), 1e-3 * scale)
        errs += [(e, k), (en, k + " [norm]")]
    errs.sort(reverse=True)
    for e, k in errs:
        # text_proj: 7e-3 on the tie-free mlm / mrc batches (measured 5.4e-3 / < 5e-3), 2e-2 with near-ties and for sap (1.1e-2
        # WITHOUT any near-tie: the reference's fp16 backward, see the module docstring)
        bound = (7e-3 if (notie and task != "sap") else 2e-2) if "text_proj" in k else 5e-3
        assert e < bound, (k, e, errs[:8])
    a, b = np.concatenate(got_all).astype(np.float64), np.concatenate(ref_all).astype(np.float64)
    cos = float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))
    assert cos > 0.999, cos


def test_backbone_positional_surface_matches_reference():
    """GlocalTextPathCMT.forward(txt_ids, txt_lens, traj_view_img_fts, ..., grid_fts, grid_map, gridmap_pos_fts=...) and
    forward_mlm(...) with the reference's positional signature (pretrain_src/model/vilmodel.py:668-673, 767-772) against
    tests/golden/pretrain_backbone_reduced.npz (the imported reference called the same way)."""
    import collections
    import inspect
    from gridmm_amd.pretrain_cmt import GlocalTextPathCMT
    from gridmm_amd.synthetic import batch_to
    want_sig = list(gen_golden.BACKBONE_ARGS) + ["target_patch_id", "gridmap_pos_fts", "return_gmap_embeds"]
    assert list(inspect.signature(GlocalTextPathCMT.forward).parameters)[1:] == want_sig
    assert list(inspect.signature(GlocalTextPathCMT.forward_mlm).parameters)[1:] == list(gen_golden.BACKBONE_ARGS) + ["gridmap_pos_fts"]
    fx = load_golden("pretrain_backbone_reduced.npz")
    model = _model(load_golden("pretrain_reduced.npz")).eval()
    with torch.no_grad():
        b = collections.defaultdict(lambda: None, batch_to(gen_golden.pretrain_batch("sap"), "cuda"))
        args = [b[k] for k in gen_golden.BACKBONE_ARGS]
        g, v, m = model.bert(*args, gridmap_pos_fts=b["gridmap_pos_fts"])
        for got, key in ((g, "sap_gmap_embeds"), (v, "sap_vp_embeds"), (m, "sap_gridmap_embeds")):
            want = torch.from_numpy(fx[key])
            assert got.shape == want.shape
            err = float((got.float().cpu() - want).abs().max())
            assert err < 3e-3 * max(1.0, float(want.abs().max())), (key, err)      # reference: fp16 grid_proj + reduction
        none_g, v2, _ = model.bert(*args, gridmap_pos_fts=b["gridmap_pos_fts"], return_gmap_embeds=False)
        assert none_g is None and torch.equal(v2, v)
        b = collections.defaultdict(lambda: None, batch_to(gen_golden.pretrain_batch("mlm"), "cuda"))
        t = model.bert.forward_mlm(*[b[k] for k in gen_golden.BACKBONE_ARGS], b["gridmap_pos_fts"])
        want = torch.from_numpy(fx["mlm_txt_embeds"])
        assert t.shape == want.shape
        assert float((t.float().cpu() - want).abs().max()) < 3e-3 * max(1.0, float(want.abs().max()))
