"""GPU: aggregation + full forward('navigation') / ('language') / ('panorama') on HIP vs the reference's
golden vectors and vs the oracle.  Tolerance: 1e-3 on logits (north star); we assert 2e-4."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden, golden_state_dict, golden_nav_batch
from oracle import navcmt_oracle as O

pytestmark = pytest.mark.gpu
LOGIT_TOL = 2e-4      # north star: 1e-3
EMBED_TOL = 5e-4


def _model(fx, dev="cuda"):
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    cfg = default_config(**json.loads(str(fx["cfg"])))
    m = GlocalTextPathNavCMT(cfg).to(dev).eval()
    sd = golden_state_dict(fx)
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    return m, sd


def _cmp(a, b, tol):
    a = a.detach().float().cpu().numpy()
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    inf = ~np.isfinite(b)
    assert np.array_equal(~np.isfinite(a), inf), "mask (-inf) placement differs"
    err = float(np.abs(a[~inf] - b[~inf]).max()) if (~inf).any() else 0.0
    assert err <= tol, err
    return err


def _to_dev(batch):
    from gridmm_amd.synthetic import batch_to
    return batch_to(batch, "cuda")


@pytest.mark.parametrize("name", ["nav_reduced.npz", "nav_reduced_obj.npz"])
def test_navigation_matches_reference_golden(name):
    fx = load_golden(name)
    model, _ = _model(fx)
    batch = _to_dev(golden_nav_batch(fx))
    outs = model("navigation", batch)
    for k in ("global_logits", "local_logits", "fused_logits", "grid_logits"):
        _cmp(outs[k], fx["out_" + k], LOGIT_TOL)
    _cmp(outs["gmap_embeds"], fx["out_gmap_embeds"], EMBED_TOL)
    _cmp(outs["vp_embeds"], fx["out_vp_embeds"], EMBED_TOL)
    if "out_obj_logits" in fx.files:
        _cmp(outs["obj_logits"], fx["out_obj_logits"], LOGIT_TOL)
    else:
        assert outs["obj_logits"] is None


def test_aggregation_stage_matches_reference_capture():
    """Cell vectors + compaction mask (vilmodel.py:793-823) incl. the mask quirk, vs the reference's
    grid_encoder input captured by a forward pre-hook."""
    from gridmm_amd import ops
    from gridmm_amd.grid_memory import pack_reference_lists
    fx = load_golden("nav_reduced.npz")
    model, sd = _model(fx)
    batch = _to_dev(golden_nav_batch(fx))
    B, L, H = batch["txt_embeds"].shape
    text_fts = ops.linear(batch["txt_embeds"], model._lin(model.text_proj, "text_proj")).f32
    slab, perm, cs = pack_reference_lists(batch["grid_fts"], batch["grid_map"])
    cells, occ, rel = ops.grid_aggregate(slab, perm, cs, ops.text_fragments(text_fts), L, want_relevance=True)
    # oracle for the un-projected stage
    cpu = golden_nav_batch(fx)
    with torch.no_grad():
        tf = O.linear(sd, "text_proj", cpu["txt_embeds"])
        for b in range(B):
            x = cpu["grid_fts"][b].float()
            w = (x @ tf[b].t()).max(-1).values
            n = x.shape[0]
            valid = cpu["grid_map"][b] >= 0
            n_valid = int(cs[b, 196])                                  # relevance comes back by sorted position
            got = torch.zeros(n)
            got[perm[b, :n_valid].long().cpu()] = rel[b, :n_valid].cpu()
            assert (got[valid] - w[valid]).abs().max() < 5e-5 * max(1.0, w.abs().max().item())
            for c in range(196):
                sel = cpu["grid_map"][b] == c
                assert bool(occ[b, c]) == bool(sel.any())
                if sel.any():
                    ref = (torch.softmax(w[sel], 0)[:, None] * x[sel]).sum(0)
                    assert (cells[b, c].cpu() - ref).abs().max() < 1e-4   # bf16x3 text_proj noise enters through softmax(w)
                else:
                    assert (cells[b, c] == 0).all()
    proj = ops.linear(cells, model._lin(model.grid_proj, "grid_proj")).f32
    gp = model.grid_pos_embeddings
    pos_emb = model._ln(gp[1], ops.linear(batch["gridmap_pos_fts"], model._lin(gp[0], "grid_pos"))).f32
    out = torch.zeros(B, 196 + 2, H, device="cuda")
    mask = torch.zeros(B, 196 + 2, dtype=torch.uint8, device="cuda")
    n_cells, cmax = ops.cells_compact(proj, pos_emb, occ, out, mask)
    C = fx["cap_grid_masks"].shape[1]
    assert int(cmax) == C
    assert np.array_equal(mask[:, :C].cpu().numpy().astype(bool), fx["cap_grid_masks"])
    assert (mask[:, C:196] == 0).all()
    assert np.abs(out[:, :C].cpu().numpy() - fx["cap_grid_map_embeds"]).max() < 2e-4


@pytest.mark.parametrize("fixture", ["text_pano_reduced.npz", "text_pano_full_b2.npz"])
def test_text_and_panorama_modes_match_reference_golden(fixture):
    """(the second fixture: the released model size, B = 2, L = 40)"""
    fx = load_golden(fixture)
    model, _ = _model(fx)
    d = lambda k: torch.from_numpy(fx[k]).cuda()
    txt = model("language", {"txt_ids": d("in_txt_ids"), "txt_masks": d("in_txt_masks")})
    valid = fx["in_txt_masks"]
    err = np.abs(txt.cpu().numpy() - fx["out_txt_embeds"])[valid].max()
    assert err < EMBED_TOL, err
    pano, pm = model("panorama", {"view_img_fts": d("in_view_img_fts"), "obj_img_fts": None, "loc_fts": d("in_loc_fts"),
                                  "nav_types": d("in_nav_types"), "view_lens": d("in_view_lens"), "obj_lens": None})
    assert np.array_equal(pm.cpu().numpy(), fx["out_pano_masks"])
    _cmp(pano, fx["out_pano_embeds"], EMBED_TOL)


def test_full_size_navigation_matches_reference_golden():
    """161 M-parameter config, B=2, N=1764/1176 points, L=40; inputs regenerated from seeds."""
    from oracle import gen_golden
    fx = load_golden("nav_full_b2.npz")
    model, _ = _model(fx)
    batch = _to_dev(gen_golden.full_b2_inputs())
    outs = model("navigation", batch)
    for k in ("global_logits", "local_logits", "fused_logits", "grid_logits"):
        _cmp(outs[k], fx["out_" + k], LOGIT_TOL)
    _cmp(outs["gmap_embeds"], fx["out_gmap_embeds"], EMBED_TOL)
    _cmp(outs["vp_embeds"], fx["out_vp_embeds"], EMBED_TOL)


def test_grid_memory_handle_equals_list_form():
    """batch['grid_memory'] (device-resident, HIP-binned) == the reference's list form of the same memory."""
    from gridmm_amd import synthetic as S
    from gridmm_amd.grid_memory import GridMemoryBatch
    fx = load_golden("nav_reduced.npz")
    model, _ = _model(fx)
    batch = _to_dev(golden_nav_batch(fx))
    B = 3
    rs = np.random.RandomState(5)
    mem = GridMemoryBatch(B, S.NATIVE, max_steps=2)
    for t in range(2):
        eps = [S.make_observations(rs, S.NATIVE, 1, feat_scale=0.35)[0] for _ in range(B)]
        mem.step(np.stack([e["depth"].reshape(-1) for e in eps]), np.stack([e["feats"] for e in eps]),
                 [(e["x"], e["y"]) for e in eps], [e["heading"] for e in eps])
    fts, gmaps, pos = mem.as_reference_obs()
    b1 = dict(batch, grid_fts=fts, grid_map=gmaps, gridmap_pos_fts=pos)
    b2 = dict(batch, grid_fts=None, grid_map=None, gridmap_pos_fts=None, grid_memory=mem)
    o1, o2 = model("navigation", b1), model("navigation", b2)
    for k in ("fused_logits", "grid_logits", "gmap_embeds"):
        a, b = o1[k], o2[k]
        f = torch.isfinite(a)
        assert torch.equal(f, torch.isfinite(b)) and (a[f] - b[f]).abs().max() < 1e-5


def _vlnce_model(fx):
    from gridmm_amd.vilmodel_ce import GlocalTextPathNavCMT, default_config
    from oracle.ref_harness import det_tensor
    m = GlocalTextPathNavCMT(default_config(**json.loads(str(fx["cfg"]))))
    names = json.loads(str(fx["param_names"]))
    assert sorted(names) == sorted(m.state_dict().keys())            # the VLN-CE module tree (no sprel_linear)
    m.load_state_dict({k: det_tensor(k, v.shape, int(fx["weight_seed"])) for k, v in m.state_dict().items()})
    return m.cuda().eval()


def test_vlnce_navigation_matches_reference_golden():
    """VLN-CE twin (gridmap/vilmodel.py:710-800): tuple batch in, fused_logits only out."""
    from oracle import gen_golden
    fx = load_golden("nav_vlnce_reduced.npz")
    model = _vlnce_model(fx)
    batch = _to_dev(golden_nav_batch(fx))
    tup = gen_golden.vlnce_nav_tuple(batch, fx["cand_lens"].tolist())
    with torch.no_grad():
        fused = model("navigation", tup)
    _cmp(fused, fx["out_fused_logits"], LOGIT_TOL)
    model.differentiable = True                                       # the autograd path gives the same logits
    fused2 = model("navigation", tup)
    assert fused2.requires_grad
    _cmp(fused2, fx["out_fused_logits"], LOGIT_TOL)
    m = torch.isfinite(fused2)
    fused2[m].sum().backward()
    assert model.grid_proj.weight.grad is not None and model.grid_sap_head.net[0].weight.grad is None


def test_vlnce_navigation_full_size_matches_reference_golden():
    """The VLN-CE twin at the released model size (B = 2, up to 1764 points of 768-D CLIP tokens in memory):
    tests/golden/nav_vlnce_full_b2.npz, inputs regenerated from the generator's seeds."""
    from oracle import gen_golden
    fx = load_golden("nav_vlnce_full_b2.npz")
    model = _vlnce_model(fx)
    batch = _to_dev(gen_golden.vlnce_full_inputs())
    with torch.no_grad():
        fused = model("navigation", gen_golden.vlnce_nav_tuple(batch, fx["cand_lens"].tolist()))
    _cmp(fused, fx["out_fused_logits"], LOGIT_TOL)


@pytest.mark.parametrize("tag", ["shared", "own"])
def test_panorama_with_object_tokens_matches_reference_golden(tag):
    """vilmodel.py:745-764: [views | objects] per panorama, objects through img_linear (REVERIE) or obj_linear."""
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    from oracle.ref_harness import det_tensor
    fx = load_golden("pano_obj_reduced.npz")
    m = GlocalTextPathNavCMT(default_config(**json.loads(str(fx["cfg_" + tag])))).cuda().eval()
    m.load_state_dict({k: det_tensor(k, v.shape, int(fx["weight_seed"])) for k, v in m.state_dict().items()})
    d = lambda k: torch.from_numpy(fx[k]).cuda()
    batch = {"view_img_fts": d("in_view_img_fts"), "obj_img_fts": d("in_obj_img_fts_" + tag), "loc_fts": d("in_loc_fts"),
             "nav_types": d("in_nav_types"), "view_lens": d("in_view_lens"), "obj_lens": d("in_obj_lens")}
    want, wmask = fx["out_pano_embeds_" + tag], fx["out_pano_masks_" + tag]
    with torch.no_grad():
        pano, pm = m("panorama", batch)
    assert np.array_equal(pm.cpu().numpy(), wmask)
    assert np.abs(pano.cpu().numpy() - want)[wmask].max() < EMBED_TOL
    m.differentiable = True
    pano2, pm2 = m("panorama", batch)
    assert pano2.requires_grad and np.abs(pano2.detach().cpu().numpy() - want)[wmask].max() < EMBED_TOL
    (pano2 * pm2.unsqueeze(-1)).sum().backward()
    g = m.img_embeddings.obj_linear.weight.grad if tag == "own" else m.img_embeddings.img_linear.weight.grad
    assert g is not None and torch.isfinite(g).all()


def test_empty_and_degenerate_grid_memories_match_oracle():
    """Edge cases of the aggregation (vilmodel.py:797-823): an episode whose points are ALL outside the map (no occupied
    cell: the sequence is the gmap nodes only), one with every point in a single cell, one with a single point, next to
    a normal one -- logits against the oracle (which follows the reference's loops literally)."""
    fx = load_golden("nav_reduced.npz")
    model, sd = _model(fx)
    cpu = golden_nav_batch(fx)
    rs = np.random.RandomState(11)
    B = cpu["txt_embeds"].shape[0]
    assert B == 3
    fts = [torch.from_numpy((rs.standard_normal((n, 768)) * 0.35).astype(np.float16)) for n in (64, 200, 1)]
    maps = [torch.full((64,), -1.0, dtype=torch.float64),          # nothing inside the 14x14 window
            torch.full((200,), 77.0, dtype=torch.float64),         # one crowded cell
            torch.tensor([195.0], dtype=torch.float64)]            # a single point in the last cell
    cpu["grid_fts"], cpu["grid_map"] = fts, maps
    with torch.no_grad():
        want = O.forward_navigation(sd, cpu)
        got = model("navigation", _to_dev(cpu))
    for k in ("global_logits", "local_logits", "fused_logits", "grid_logits"):
        _cmp(got[k], want[k].numpy(), LOGIT_TOL)
    _cmp(got["gmap_embeds"], want["gmap_embeds"].numpy(), EMBED_TOL)


@pytest.mark.parametrize("seed,geom_name,T,long", [(1, "NATIVE", 3, False), (2, "NATIVE", 6, False), (3, "BASELINE", 2, False),
                                                   (4, "NATIVE", 1, False), (5, "NATIVE", 2, True)])
def test_step_sequence_matches_oracle_over_seeds(seed, geom_name, T, long):
    """fill_gridmap over T observations + forward('navigation') on the device-resident memory vs the oracle's literal
    loops, fresh random episodes per seed: cell ids bit-exact at every step, logits within LOGIT_TOL.  NATIVE runs the
    two-pass D = 768 aggregation (relevance pass + accumulation pass), BASELINE the single pipelined kernel."""
    from gridmm_amd import synthetic as S
    from gridmm_amd.grid_memory import GridMemoryBatch
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    from oracle import gridmap_oracle as G
    from oracle.ref_harness import det_tensor
    geom, og = getattr(S, geom_name), getattr(G, geom_name)
    cfg = default_config(num_l_layers=1, num_pano_layers=1, num_x_layers=2, intermediate_size=256, vocab_size=1000,
                         grid_feat_size=geom.feat_dim)
    model = GlocalTextPathNavCMT(cfg).eval()
    sd = {k: (det_tensor(k, v.shape, seed) if v.dtype.is_floating_point else v) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    model.cuda()
    rs = np.random.RandomState(100 + seed)
    B = 3
    mem = GridMemoryBatch(B, geom, max_steps=T, device="cuda")
    oracles = [G.GridMemory(og) for _ in range(B)]
    eps = [S.make_observations(rs, geom, T, feat_scale=0.35) for _ in range(B)]
    for t in range(T):
        mem.step(np.stack([e[t]["depth"].reshape(-1) for e in eps]), np.stack([e[t]["feats"] for e in eps]),
                 [(e[t]["x"], e[t]["y"]) for e in eps], [e[t]["heading"] for e in eps])
        ref = [oracles[b].step(eps[b][t]["depth"], eps[b][t]["feats"], eps[b][t]["x"], eps[b][t]["y"],
                               eps[b][t]["heading"]) for b in range(B)]
        for b in range(B):
            n = ref[b][1].shape[0]
            assert np.array_equal(mem.cell_id[b, :n].cpu().numpy(), ref[b][1].astype(np.int16)), (t, b)
    if long:   # RxR-sized sequences (scripts/run_rxr.sh: max_instr_len 250; rxr_pretrain.json: 300) and a long topological
        # map: 196 + 90 + 300 = 586 keys in the local encoder's context (past the 512 keys of the round-2 attention kernel)
        batch = S.make_nav_batch(rs, B, L=300, G=90, n_visited=40, V1=37, n_cand=5, min_len=120)
    else:
        batch = S.make_nav_batch(rs, B, L=40 + 10 * seed, G=8, n_visited=3, V1=10, n_cand=3, min_len=8)
    cpu = dict(batch, grid_fts=[torch.from_numpy(r[0]) for r in ref], grid_map=[torch.from_numpy(r[1]) for r in ref],
               gridmap_pos_fts=torch.from_numpy(np.stack([r[2] for r in ref])))
    with torch.no_grad():
        want = O.forward_navigation(sd, cpu)
        got = model("navigation", dict(S.batch_to(batch, "cuda"), grid_memory=mem, grid_fts=None, grid_map=None,
                                       gridmap_pos_fts=None))
    for k in ("global_logits", "local_logits", "fused_logits", "grid_logits"):
        _cmp(got[k], want[k].numpy(), LOGIT_TOL)
