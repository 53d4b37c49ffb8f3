"""Generate tests/golden/*.npz by IMPORTING the reference (build container only).

TEST INFRASTRUCTURE.  Run:  python -m oracle.gen_golden
Needs /root/reference; the produced fixtures are data only (inputs + the
reference's outputs) and travel to the GPU box, the reference does not.

Fixtures:
  fill_gridmap_native.npz   EnvBatch.getGlobalMap sequences (env.py:267-374): random walk,
                            all-zero depth first step, negative coords, arbitrary headings
  nav_reduced.npz           forward('navigation') (vilmodel.py:782-918), reduced config
                            (1/1/1 layers, FFN 64, vocab 2000), B=3 ragged N; all outputs +
                            the grid_encoder input/mask captured by a forward pre-hook
  nav_reduced_obj.npz       same with obj_feat_size=768 (og_head + vp_obj_masks)
  nav_full_b2.npz           full-size config (9/2/4 layers, 161 M params), B=2, t=3;
                            inputs regenerated from seeds -> only outputs stored
  text_pano_reduced.npz     forward('language') / forward('panorama') reduced config
Weights are never stored: both sides regenerate them with ref_harness.det_tensor.
"""
import json
import math
import os
import sys

import numpy as np
import torch

from . import ref_harness as R
from . import gridmap_oracle as G

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gridmm_amd import synthetic as S  # noqa: E402  (input generators only)

OUT = os.environ.get("GRIDMM_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
REDUCED = dict(num_l_layers=1, num_pano_layers=1, num_x_layers=1, intermediate_size=64, vocab_size=2000)


def _versions():
    import transformers
    return json.dumps({"numpy": np.__version__, "torch": torch.__version__,
                       "transformers": transformers.__version__})


def _full_depth(depth_s):
    """Embed sampled (12,49) depth into a (36,128,128,1) map at the reference's sample indices."""
    full = np.zeros((36, 128, 128, 1), np.uint16)
    idx = G.NATIVE.sample_index()
    for v in range(12):
        full[12 + v][np.ix_(idx, idx)] = depth_s[v].reshape(7, 7, 1)
    return full


def gen_fill_gridmap():
    rs = np.random.RandomState(1234)
    episodes = []
    # A: random walk, 4 steps;  B: all-zero depth first step;  C: far-negative coords + odd headings
    for name, steps in (("A", 4), ("B", 3), ("C", 4), ("D", 1)):
        obs = S.make_observations(rs, S.NATIVE, steps, feat_scale=1.0)
        if name == "B":
            obs[0]["depth"][:] = 0
        if name == "C":
            for o in obs:
                o["x"] -= 137.25
                o["y"] -= 61.125
                o["heading"] = float(rs.uniform(-7, 7))
        if name == "D":
            obs[0]["depth"][:, ::2] = 0
        episodes.append(obs)
    depth_db, clip_db, info = {}, {}, {}
    for e, obs in enumerate(episodes):
        for t, o in enumerate(obs):
            key = "s%d_v%d" % (e, t)
            depth_db[key] = _full_depth(o["depth"])
            clip = np.zeros((12, 50, 768), np.float16)
            clip[:, 1:] = o["feats"].reshape(12, 49, 768)
            clip_db[key] = clip
            info[key] = {"x": o["x"], "y": o["y"]}
    env = R.RefGridEnv(len(episodes), depth_db, clip_db, info)
    out = {"versions": _versions(), "n_episodes": len(episodes)}
    for e, obs in enumerate(episodes):
        out["e%d_steps" % e] = len(obs)
        for t, o in enumerate(obs):
            sem, gmap, pos = env.step(e, "s%d" % e, "v%d" % t, o["heading"])
            assert sem.shape[0] == 588 * (t + 1)
            assert np.array_equal(sem[-588:], o["feats"])
            p = "e%d_t%d_" % (e, t)
            out[p + "depth"] = o["depth"]
            out[p + "pose"] = np.array([o["x"], o["y"], o["heading"]], np.float64)
            out[p + "grid_map"] = gmap.astype(np.int16)
            out[p + "pos_fts"] = pos.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "fill_gridmap_native.npz"), **out)
    print("fill_gridmap_native: ok")


def _nav_inputs(seed, B, Ns, L, G_, V1, n_cand, n_visited, with_obj=False, feat_scale=0.35):
    rs = np.random.RandomState(seed)
    batch = S.make_nav_batch(rs, B, L=L, G=G_, n_visited=n_visited, V1=V1, n_cand=n_cand,
                             min_len=max(2, L // 3), with_obj=with_obj)
    grid_fts, grid_map, pos = [], [], []
    for b in range(B):
        n = Ns[b]
        grid_fts.append(torch.from_numpy((rs.standard_normal((n, 768)) * feat_scale).astype(np.float16)))
        gm = rs.randint(-1, 196, size=n).astype(np.float64)
        if b == 1:  # sparse occupancy: few cells -> exercises the compaction-mask quirk
            gm = rs.choice([-1, 3, 17, 18, 95, 96, 150, 195], size=n).astype(np.float64)
        grid_map.append(torch.from_numpy(gm))
        pos.append(torch.from_numpy(G.gridmap_pos_fts(np.float32(rs.uniform(2, 9)))))
    batch["grid_fts"], batch["grid_map"] = grid_fts, grid_map
    batch["gridmap_pos_fts"] = torch.stack(pos, 0)
    return batch


def _pack_batch(out, batch):
    for k, v in batch.items():
        if torch.is_tensor(v):
            out["in_" + k] = v.numpy()
        elif k in ("grid_fts", "grid_map"):
            for b, t in enumerate(v):
                out["in_%s_%d" % (k, b)] = t.numpy()
        elif k in ("gmap_vpids", "vp_cand_vpids"):
            out["in_" + k] = json.dumps(v)


def _run_nav(model, batch):
    cap = {}

    def hook(mod, args, kwargs):
        cap["map_embeds"] = args[0].detach().clone()
        cap["kpm"] = kwargs["src_key_padding_mask"].detach().clone()

    h = model.grid_encoder.register_forward_pre_hook(hook, with_kwargs=True)
    with torch.no_grad():
        outs = model("navigation", batch)
    h.remove()
    return outs, cap


def gen_nav_reduced(with_obj):
    torch.set_num_threads(1)
    over = dict(REDUCED)
    if with_obj:
        over["obj_feat_size"] = 768
    model = R.build_ref_model(seed=7, **over)
    batch = _nav_inputs(seed=99 + int(with_obj), B=3, Ns=[230, 170, 96], L=12, G_=7, V1=9, n_cand=3,
                        n_visited=2, with_obj=with_obj)
    outs, cap = _run_nav(model, batch)
    out = {"versions": _versions(), "weight_seed": 7, "cfg": json.dumps(over),
           "param_names": json.dumps([k for k in model.state_dict()]),
           "param_shapes": json.dumps([list(v.shape) for v in model.state_dict().values()])}
    _pack_batch(out, batch)
    for k, v in outs.items():
        if v is not None:
            out["out_" + k] = v.numpy()
    C = cap["map_embeds"].shape[1] - batch["gmap_masks"].shape[1]
    out["cap_grid_map_embeds"] = cap["map_embeds"][:, :C].numpy()
    out["cap_grid_masks"] = cap["kpm"][:, :C].logical_not().numpy()
    name = "nav_reduced_obj.npz" if with_obj else "nav_reduced.npz"
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "ok  Cmax=%d" % C, {k: tuple(v.shape) for k, v in outs.items() if v is not None})


def full_b2_inputs():
    """Inputs of nav_full_b2 (regenerated identically by the tests from these seeds)."""
    return _nav_inputs(seed=4242, B=2, Ns=[1764, 1176], L=40, G_=12, V1=37, n_cand=4, n_visited=4)


def gen_nav_full():
    torch.set_num_threads(4)
    model = R.build_ref_model(seed=3)
    batch = full_b2_inputs()
    outs, cap = _run_nav(model, batch)
    out = {"versions": _versions(), "weight_seed": 3, "cfg": json.dumps({}),
           "param_names": json.dumps([k for k in model.state_dict()]),
           "param_shapes": json.dumps([list(v.shape) for v in model.state_dict().values()])}
    for k, v in outs.items():
        if v is not None:
            out["out_" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "nav_full_b2.npz"), **out)
    print("nav_full_b2 ok", {k: float(v[torch.isfinite(v)].abs().max()) for k, v in outs.items() if v is not None})


def gen_text_pano(full=False):
    """forward('language') / forward('panorama').  full: the released model size (9 language layers, 2 panorama layers,
    30 522-word vocabulary), B = 2, L = 40."""
    torch.set_num_threads(4 if full else 1)
    cfg = {} if full else REDUCED
    model = R.build_ref_model(seed=7, **cfg)
    rs = np.random.RandomState(6 if full else 5)
    B, L = (2, 40) if full else (3, 14)
    lens = np.array([40, 23]) if full else np.array([14, 9, 5])
    txt_ids = rs.randint(1, 30000 if full else 2000, size=(B, L)).astype(np.int64) * (np.arange(L)[None] < lens[:, None])
    txt_masks = np.arange(L)[None] < lens[:, None]
    view = rs.standard_normal((B, 36, 768)).astype(np.float32)
    loc = rs.uniform(-1, 1, size=(B, 36, 7)).astype(np.float32)
    nav_types = (rs.rand(B, 36) < 0.15).astype(np.int64)
    view_lens = np.full(B, 36, np.int64)
    with torch.no_grad():
        txt = model("language", {"txt_ids": torch.from_numpy(txt_ids), "txt_masks": torch.from_numpy(txt_masks)})
        pano, pmask = model("panorama", {
            "view_img_fts": torch.from_numpy(view), "obj_img_fts": None, "loc_fts": torch.from_numpy(loc),
            "nav_types": torch.from_numpy(nav_types), "view_lens": torch.from_numpy(view_lens), "obj_lens": None})
    np.savez_compressed(
        os.path.join(OUT, "text_pano_full_b2.npz" if full else "text_pano_reduced.npz"), versions=_versions(), weight_seed=7,
        cfg=json.dumps(cfg),
        param_names=json.dumps([k for k in model.state_dict()]),
        param_shapes=json.dumps([list(v.shape) for v in model.state_dict().values()]),
        in_txt_ids=txt_ids, in_txt_masks=txt_masks, in_view_img_fts=view, in_loc_fts=loc,
        in_nav_types=nav_types, in_view_lens=view_lens, out_txt_embeds=txt.numpy(),
        out_pano_embeds=pano.numpy(), out_pano_masks=pmask.numpy())
    print("text_pano_full_b2 ok" if full else "text_pano_reduced ok")


def gen_pano_obj():
    """forward('panorama') with object tokens (vilmodel.py:745-764), for both embeddings of the objects:
    obj_feat_size == image_feat_size (shared img_linear, REVERIE) and != (own obj_linear / obj_layer_norm)."""
    torch.set_num_threads(1)
    out = {"versions": _versions(), "weight_seed": 7}
    rs = np.random.RandomState(15)
    B = 3
    view_lens, obj_lens = np.array([36, 33, 36], np.int64), np.array([5, 0, 9], np.int64)
    P = int((view_lens + obj_lens).max())
    view = rs.standard_normal((B, 36, 768)).astype(np.float32)
    loc = rs.uniform(-1, 1, size=(B, P, 7)).astype(np.float32)
    nav_types = np.zeros((B, P), np.int64)
    for b in range(B):
        nav_types[b, :3] = 1
        nav_types[b, view_lens[b]:view_lens[b] + obj_lens[b]] = 2
    out.update(in_view_img_fts=view, in_loc_fts=loc, in_nav_types=nav_types, in_view_lens=view_lens, in_obj_lens=obj_lens)
    for tag, osz in (("shared", 768), ("own", 64)):
        over = dict(REDUCED, obj_feat_size=osz)
        model = R.build_ref_model(seed=7, **over)
        obj = rs.standard_normal((B, int(obj_lens.max()), osz)).astype(np.float32)
        with torch.no_grad():
            pano, pmask = model("panorama", {
                "view_img_fts": torch.from_numpy(view), "obj_img_fts": torch.from_numpy(obj),
                "loc_fts": torch.from_numpy(loc), "nav_types": torch.from_numpy(nav_types),
                "view_lens": torch.from_numpy(view_lens), "obj_lens": torch.from_numpy(obj_lens)})
        out["cfg_" + tag] = json.dumps(over)
        out["in_obj_img_fts_" + tag] = obj
        out["out_pano_embeds_" + tag] = pano.numpy()
        out["out_pano_masks_" + tag] = pmask.numpy()
    np.savez_compressed(os.path.join(OUT, "pano_obj_reduced.npz"), **out)
    print("pano_obj_reduced ok", pano.shape)


def gen_fill_gridmap_vlnce():
    """VLN-CE twin: GridMap.getGlobalMap (Policy_ViewSelection_GridMap.py:689-825), R2R-CE and RxR-CE constants."""
    out = {"versions": _versions()}
    idx = G.VLNCE_R2R.sample_index()
    for name, geom, md in (("r2r", G.VLNCE_R2R, 25), ("rxr", G.VLNCE_RXR, 40)):
        rs = np.random.RandomState(77 if name == "r2r" else 78)
        env = R.RefVlnceGridEnv(1, name.replace("r2r", "R2R").replace("rxr", "RxR"), md)
        T = 4
        out[name + "_steps"] = T
        for t in range(T):
            ds = rs.uniform(0.0, 6.0, size=(12, 49)).astype(np.float32)
            ds[rs.rand(12, 49) < (1.0 if (name == "rxr" and t == 0) else 0.12)] = 0.0   # RxR ep: all-invalid first step
            full = np.zeros((12, 256, 256), np.float32)
            for v in range(12):
                full[v][np.ix_(idx, idx)] = ds[v].reshape(7, 7)
            ft = np.zeros((12, 50, 768), np.float16)
            pos = {"x": float(rs.uniform(-8, 8)), "y": float(rs.uniform(-8, 8))}
            h = float(rs.uniform(-6.5, 6.5))
            fts, gmap, pf = env.step(0, pos, h, full, ft)
            p = "%s_t%d_" % (name, t)
            out[p + "depth"] = ds
            out[p + "pose"] = np.array([pos["x"], pos["y"], h], np.float64)
            out[p + "grid_map"] = gmap.astype(np.int16)
            out[p + "pos_fts"] = pf.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "fill_gridmap_vlnce.npz"), **out)
    print("fill_gridmap_vlnce: ok")


ROLLOUT = dict(batch_size=3, n_scans=2, n_episodes=3, seed=12, max_action_len=6)


def make_rollout_agent(vln_bert, device="cpu", grid_memory=None):
    """The scripted synthetic episodes of rollout_reduced.npz (shared by the generator and the tests)."""
    from gridmm_amd.agent import GMapNavAgent, default_args
    from gridmm_amd.sim_env import SyntheticNavEnv
    from .adapters import OracleGridMemory
    r = ROLLOUT
    mem = grid_memory if grid_memory is not None else OracleGridMemory(r["batch_size"])
    env = SyntheticNavEnv(r["batch_size"], mem, n_scans=r["n_scans"], n_episodes=r["n_episodes"], seed=r["seed"])
    agent = GMapNavAgent(default_args(max_action_len=r["max_action_len"]), env, vln_bert, device=device)
    agent.feedback = "argmax"
    agent.trace = []
    return agent


def gen_rollout(full=False):
    """GMapNavAgent.rollout (agent.py:268-451) driven with the REFERENCE model: per-step logits + actions.
    full: the released model size inside the loop (rollout_full.npz)."""
    import collections
    torch.set_num_threads(4 if full else 1)
    cfg = {} if full else REDUCED
    wseed = 7          # (full size, episode seed 12: 6 steps with a 2.7e-3 argmax margin; weight seeds 3 / 5 / 9 / 11 / 13 stop at
    # step 0 or leave margins below the fixture's 2e-3 bar: oracle/search_rollout_seeds.py)
    model = R.build_ref_model(seed=wseed, **cfg)

    def ref_bert(mode, batch):
        with torch.no_grad():
            return model(mode, collections.defaultdict(lambda: None, batch))

    agent = make_rollout_agent(ref_bert)
    agent.fast_collate = False        # the per-episode restatement of the reference's collation feeds the reference model
    traj = agent.rollout()
    out = {"versions": _versions(), "weight_seed": wseed, "cfg": json.dumps(cfg),
           "param_names": json.dumps([k for k in model.state_dict()]),
           "param_shapes": json.dumps([list(v.shape) for v in model.state_dict().values()]),
           "n_steps": len(agent.trace), "traj": json.dumps([t["path"] for t in traj])}
    margin = 1e9
    for st in agent.trace:
        t = st["t"]
        fl = st["nav_outs"]["fused_logits"]
        out["t%d_fused" % t] = fl.numpy()
        out["t%d_local" % t] = st["nav_outs"]["local_logits"].numpy()
        out["t%d_global" % t] = st["nav_outs"]["global_logits"].numpy()
        out["t%d_grid" % t] = st["nav_outs"]["grid_logits"].numpy()
        out["t%d_a" % t] = st["a_t"]
        out["t%d_ended" % t] = st["ended"]
        top2 = torch.topk(torch.nan_to_num(fl, neginf=-1e9), 2, dim=1).values
        margin = min(margin, float((top2[:, 0] - top2[:, 1])[~torch.from_numpy(st["ended"])].min()) if (~st["ended"]).any() else margin)
    assert margin > 2e-3, "argmax margin %.2e too small for a robust action fixture; change ROLLOUT seed" % margin
    np.savez_compressed(os.path.join(OUT, "rollout_full.npz" if full else "rollout_reduced.npz"), **out)
    print(("rollout_full" if full else "rollout_reduced") + " ok: steps=%d min argmax margin=%.3e paths=%s" % (len(agent.trace), margin, out["traj"][:120]))


VLNCE_NAV_CAND_LENS = [4, 3, 4]


def vlnce_nav_tuple(batch, cand_lens=VLNCE_NAV_CAND_LENS):
    """The positional `navigation` batch of the VLN-CE model (gridmap/vilmodel.py:710-713, 815-818) from the dict form."""
    return (batch["txt_embeds"], batch["txt_masks"], batch["gmap_img_embeds"], batch["gmap_step_ids"],
            batch["gmap_pos_fts"], batch["gmap_masks"], batch["vp_img_embeds"], batch["vp_pos_fts"], batch["vp_masks"],
            batch["vp_nav_masks"], batch["grid_fts"], batch["grid_map"], batch["gridmap_pos_fts"], list(cand_lens))


def vlnce_full_inputs():
    """Inputs of nav_vlnce_full_b2 (regenerated identically by the tests from these seeds): two episodes with 1-3
    observations of 12 views x 49 patches in memory."""
    return _nav_inputs(seed=654, B=2, Ns=[1764, 588], L=40, G_=12, V1=20, n_cand=3, n_visited=3)


VLNCE_FULL_CAND_LENS = [4, 3]


def gen_nav_vlnce(full=False):
    """VLN-CE GlocalTextPathNavCMT.forward('navigation', tuple) (gridmap/vilmodel.py:710-800): fused logits only.
    full: the released model size (no reduction), inputs regenerated by the tests from seeds."""
    torch.set_num_threads(4 if full else 1)
    cfg = {} if full else REDUCED
    model = R.build_ref_vlnce_model(seed=7, **cfg)
    batch = vlnce_full_inputs() if full else _nav_inputs(seed=321, B=3, Ns=[200, 150, 90], L=12, G_=7, V1=9, n_cand=3, n_visited=2)
    cand = VLNCE_FULL_CAND_LENS if full else VLNCE_NAV_CAND_LENS
    with torch.no_grad():
        fused = model("navigation", vlnce_nav_tuple(batch, cand))
    out = {"versions": _versions(), "weight_seed": 7, "cfg": json.dumps(cfg),
           "param_names": json.dumps([k for k in model.state_dict()]),
           "cand_lens": np.array(cand)}
    if not full:
        _pack_batch(out, batch)
    out["out_fused_logits"] = fused.numpy()
    name = "nav_vlnce_full_b2.npz" if full else "nav_vlnce_reduced.npz"
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "ok", tuple(fused.shape))


# mrc: RegionClassification holds a ReLU (pretrain_cmt.py:15-18); the batch seeds below are the ones (of 400 tried,
# oracle/search_pretrain_seeds.py) whose pre-activations all clear the gate by >= 1.05e-3 (views) / 6.6e-4 (views + objects),
# so that no gate can flip between two correct implementations and the mrc gradients are pinned elementwise like the rest
PRETRAIN_SEEDS = {"mlm": 11, "mrc": 371, "sap": 13}
PRETRAIN_SEEDS_OBJ = {"mrc": 147}
PRETRAIN_FULL_SEEDS = {"mlm": 211, "mrc": 212, "sap": 213}      # pretrain_full_b2.npz (unchanged since round 2)
GRAD_SAMPLES = 48


PRETRAIN_OBJ = dict(obj_feat_size=768, obj_prob_size=30, pretrain_tasks=["mrc", "sap", "og"])   # REVERIE-style objects


def pretrain_batch(task, with_obj=False):
    """The batches of pretrain_reduced{,_obj}.npz, regenerated identically by the tests (inputs are not stored)."""
    if with_obj:
        return S.make_pretrain_batch(np.random.RandomState(PRETRAIN_SEEDS_OBJ.get(task, PRETRAIN_SEEDS.get(task, 14)) + 100), 3,
                                     task, with_obj=True,
                                     obj_feat_size=PRETRAIN_OBJ["obj_feat_size"], obj_prob_size=PRETRAIN_OBJ["obj_prob_size"])
    return S.make_pretrain_batch(np.random.RandomState(PRETRAIN_SEEDS[task]), 3, task)


# the released pre-training configuration (config/r2r_model_config.json + pretrain_cmt.py defaults): full depth and width
PRETRAIN_FULL = dict(num_l_layers=9, num_pano_layers=2, num_x_layers=4, intermediate_size=3072, vocab_size=30522,
                     image_prob_size=1000)


def pretrain_full_batch(task):
    """The batches of pretrain_full_b2.npz (B = 2, L = 40, up to 4 steps, 300-900 grid points), regenerated by the tests."""
    return S.make_pretrain_batch(np.random.RandomState(PRETRAIN_FULL_SEEDS[task]), 2, task, max_steps=4, L=40, vocab=30522,
                                 image_prob_size=1000, n_pts=(300, 900))


# pretrain_reduced_notie.npz: small grid memories (40-64 points per episode) from batch seeds (oracle/search_relevance_ties.py) for
# which NO point's two best instruction tokens are closer than 4x the difference between the reference's fp16 relevance product and
# the same product with un-rounded text features: the arg-max routing of the relevance maximum (pretrain_src/model/vilmodel.py:
# 685-690) is then the same for every correct implementation, and text_proj's gradient can be pinned like any other parameter's
# (mlm: best of seeds 1000-1299 by min top-2 gap / discrepancy; mrc: best of 1300-2400 that ALSO clears the ReLU gate of
# RegionClassification by 7.8e-4, see PRETRAIN_SEEDS above; sap: best of 1000-4999 whose ClsPrediction ReLU pre-activations
# (pretrain_cmt.py:24-36: ~10^5 values per batch) ALSO all clear zero by 6.2e-5 -- those heads run in fp32 on both sides, so
# a gate can only flip inside the ~1e-5 difference of two fp32 implementations)
PRETRAIN_NOTIE_SEEDS = {"mlm": 1100, "mrc": 2331, "sap": 1871}


def pretrain_notie_batch(task):
    return S.make_pretrain_batch(np.random.RandomState(PRETRAIN_NOTIE_SEEDS[task]), 3, task, n_pts=(40, 64))


def grad_sample_index(name, numel):
    """Seeded positions at which a parameter's gradient is recorded."""
    import zlib
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return rs.randint(0, numel, size=min(GRAD_SAMPLES, numel))


def gen_pretrain(with_obj=False, full=False, notie=False):
    """GlocalTextPathCMTPreTraining.forward(batch, task) (pretrain_cmt.py:71-321) in train-step form
    (train_r2r.py:245-262): per-sample loss vectors, then loss.mean().backward(): per-parameter gradient norm,
    seeded samples of every gradient, and the set of parameters that received none.
    with_obj: object tokens in every panorama (REVERIE-style), tasks mrc (view + object branches) / sap / og."""
    torch.set_num_threads(1)
    over = dict(PRETRAIN_OBJ) if with_obj else (dict(PRETRAIN_FULL) if full else {})
    model = R.build_ref_pretrain_model(seed=9, **over).train()     # dropout probs are 0 in these configs
    out = {"versions": _versions(), "weight_seed": 9, "cfg": json.dumps(dict(R.PRETRAIN_REDUCED, **over)),
           "param_names": json.dumps([k for k, _ in model.named_parameters()]),
           "param_dtypes": json.dumps({k: str(v.dtype) for k, v in model.state_dict().items()})}
    for task in (("mrc", "sap", "og") if with_obj else ("mlm", "mrc", "sap")):
        batch = pretrain_notie_batch(task) if notie else (pretrain_full_batch(task) if full else pretrain_batch(task, with_obj))
        from oracle.search_relevance_ties import relevance_ties
        ties, pts, gap, disc = relevance_ties(model, batch, task)
        out["relevance_ties_" + task] = np.array([ties, pts], np.int64)              # near-tie points / points of the batch
        out["relevance_gap_" + task] = np.array([gap, disc], np.float32)             # min top-2 gap, max product discrepancy
        if notie:
            assert ties == 0, (task, ties, pts)
        model.zero_grad()
        # the reference's backward through the fp16 grid path (grid_proj output, softmax weights, per-cell sums are half tensors,
        # pretrain_src/model/vilmodel.py:690-703): how much of d(loss)/d(grid_proj output) lies in fp16's subnormal range -- the
        # precision text_proj's gradient inherits, whatever the arg-max routing does
        sub = []

        def _fp16_grad_stats(mod, gin, gout):
            a = gout[0].detach().float().abs()
            nz = a[a > 0]
            sub.append([float(a.max()), float(nz.median()) if nz.numel() else 0.0, float((a == 0).float().mean()),
                        float(((a > 0) & (a < 6.1e-5)).float().mean())])
        import warnings
        hk = model.bert.grid_proj.register_full_backward_hook(_fp16_grad_stats)
        loss = model(batch, task=task, compute_loss=True)
        out["loss_" + task] = loss.detach().float().numpy()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            loss.mean().backward()
        hk.remove()
        sub = np.array(sub, np.float64)      # per episode: max |g|, median nonzero |g|, share exactly zero, share subnormal
        out["fp16_grad_" + task] = np.array([sub[:, 0].max(), np.median(sub[:, 1]), sub[:, 2].mean(), sub[:, 3].mean()], np.float32)
        names, norms, samples = [], [], []
        for k, p in model.named_parameters():
            if p.grad is None:
                continue
            g = p.grad.detach().float().reshape(-1)
            names.append(k)
            norms.append(float(g.norm()))
            samples.append(g[torch.from_numpy(grad_sample_index(k, g.numel()))].numpy())
        out["grad_names_" + task] = json.dumps(names)
        out["grad_norms_" + task] = np.array(norms, np.float32)
        out["grad_samples_" + task] = np.concatenate(samples).astype(np.float32)
        print(task, "loss", out["loss_" + task], "params with grad", len(names))
    name = "pretrain_full_b2.npz" if full else ("pretrain_reduced_obj.npz" if with_obj else "pretrain_reduced.npz")
    if notie:
        name = "pretrain_reduced_notie.npz"
    np.savez_compressed(os.path.join(OUT, name), **out)


TOPO = dict(n_nodes=24, n_steps=20, seed=5, max_nodes=24)


def topo_walk_inputs():
    """Scripted walk over a random geometric graph (shared by the generator and tests/test_topo_map.py)."""
    rs = np.random.RandomState(TOPO["seed"])
    n = TOPO["n_nodes"]
    pos = np.concatenate([rs.uniform(-10, 10, (n, 2)), rs.uniform(-1, 1, (n, 1))], 1)
    d = np.sqrt(((pos[:, None] - pos[None]) ** 2).sum(-1))
    adj = np.zeros((n, n), dtype=bool)
    for i in range(n):
        for j in np.argsort(d[i])[1:4]:
            adj[i, j] = adj[j, i] = True
    walk, cur, seen = [], 0, set()
    for _ in range(TOPO["n_steps"]):
        walk.append(cur)
        seen.add(cur)
        nb = np.nonzero(adj[cur])[0]
        fresh = [j for j in nb if j not in seen]
        cur = int((fresh or list(nb))[rs.randint(len(fresh or nb))])
    return dict(pos=pos, adj=adj, walk=np.array(walk), heading=rs.uniform(0, 2 * np.pi, len(walk)),
                elevation=rs.uniform(-0.5, 0.5, len(walk)), embeds=rs.randn(len(walk), 5, 8).astype(np.float32))


def topo_observation(inp, t):
    i = int(inp["walk"][t])
    return {"viewpoint": "vp%02d" % i, "position": tuple(inp["pos"][i]),
            "candidate": [{"viewpointId": "vp%02d" % j, "position": tuple(inp["pos"][j])}
                          for j in np.nonzero(inp["adj"][i])[0]]}


def gen_topo_map():
    """map_nav_src/models/graph_utils.py:43-151 (FloydGraph + GraphMap) driven over the scripted walk: after every
    step the pair distances, hop counts, routes from the current node, visited flags, position features and node
    embedding means -> tests/golden/topo_map.npz (pins gridmm_amd/graph_utils.TopoMap)."""
    R.install_shims()
    R.use_tree(R.REF_NAV)
    from models import graph_utils as G       # the reference's module
    inp = topo_walk_inputs()
    N, T = TOPO["max_nodes"], len(inp["walk"])
    gm = G.GraphMap("vp%02d" % inp["walk"][0])
    out = {"versions": _versions()}
    out.update({"in_" + k: v for k, v in inp.items()})
    dist = np.zeros((T, N, N)); hops = np.full((T, N, N), -1, dtype=np.int64); route = np.full((T, N, N), -1, dtype=np.int64)
    order = np.full((T, N), -1, dtype=np.int64); visited = np.zeros((T, N), dtype=bool)
    pos_fts = np.zeros((T, N + 1, 7), dtype=np.float32); emb = np.zeros((T, N, 8), dtype=np.float32)
    for t in range(T):
        ob = topo_observation(inp, t)
        gm.update_graph(ob)
        cur = ob["viewpoint"]
        gm.update_node_embed(cur, torch.from_numpy(inp["embeds"][t, 0]), rewrite=True)
        for c, cc in enumerate(ob["candidate"]):
            if not gm.graph.visited(cc["viewpointId"]):
                gm.update_node_embed(cc["viewpointId"], torch.from_numpy(inp["embeds"][t, 1 + c]))
        names = list(gm.node_positions.keys())
        order[t, :len(names)] = [int(v[2:]) for v in names]
        for a, va in enumerate(names):
            visited[t, a] = gm.graph.visited(va)
            emb[t, a] = gm.get_node_embed(va).numpy()
            for b, vb in enumerate(names):
                dist[t, a, b] = gm.graph.distance(va, vb)
                hops[t, a, b] = len(gm.graph.path(va, vb))
            r = gm.graph.path(cur, va)
            route[t, a, :len(r)] = [int(v[2:]) for v in r]
        pos_fts[t, :len(names) + 1] = gm.get_pos_fts(cur, [None] + names, inp["heading"][t], inp["elevation"][t])
    out.update(dist=dist, hops=hops, route=route, order=order, visited=visited, pos_fts=pos_fts, emb=emb)
    np.savez_compressed(os.path.join(OUT, "topo_map.npz"), **out)
    print("topo_map: ok,", int((order[-1] >= 0).sum()), "nodes after", T, "steps")


OPTIM = dict(steps=7, learning_rate=3e-3, warmup_steps=3, num_train_steps=9, weight_decay=0.01, betas=(0.9, 0.98),
             grad_norm=2.0)


class OptimToy(torch.nn.Module):
    """Parameter NAMES matter (optim/misc.py:14: no decay on 'bias', 'LayerNorm.bias', 'LayerNorm.weight')."""

    def __init__(self):
        super().__init__()
        self.dense = torch.nn.Linear(24, 16)
        self.LayerNorm = torch.nn.LayerNorm(16)
        self.emb = torch.nn.Embedding(10, 16)
        self.grid_proj = torch.nn.Linear(16, 8, bias=False).half()   # the pre-training twin keeps this one in fp16
        self.unused = torch.nn.Linear(4, 4)                          # never receives a gradient


def optim_toy_grad(name, shape, step):
    """Deterministic gradients: alternating large / tiny scales so that clipping is active on some steps only, and
    |g| ~ 2e-4 entries whose squares underflow an fp16 second-moment state."""
    g = R.det_tensor("g%d.%s" % (step, name), tuple(shape), 1)
    return g * (6.0 if step % 2 == 0 else 2e-4 if step == 3 else 0.3)


def gen_optim():
    """pretrain_src/optim/adamw.py:56-112 + sched.py:17-30 + misc.py:12-37 driven as train_r2r.py:266-296 does
    (lr schedule -> clip_grad_norm_ -> step) -> tests/golden/optim_reduced.npz (pins gridmm_amd/optim.py + optim.hip)."""
    R.use_tree(R.REF_PRETRAIN)
    from optim.misc import build_optimizer          # the reference's modules
    from optim.sched import get_lr_sched
    o = OPTIM
    opts = R._AttrDict(optim="adamw", learning_rate=o["learning_rate"], betas=list(o["betas"]), weight_decay=o["weight_decay"],
                       warmup_steps=o["warmup_steps"], num_train_steps=o["num_train_steps"])
    torch.manual_seed(0)
    model = OptimToy()
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(R.det_tensor("p." + n, tuple(p.shape), 1).to(p.dtype))
    opt = build_optimizer(model, opts)
    out = {"versions": _versions(), "cfg": json.dumps(o), "names": json.dumps([n for n, _ in model.named_parameters()]),
           "decay": json.dumps([[n for n, p in model.named_parameters() if any(p is q for q in g["params"])] for g in opt.param_groups])}
    for n, p in model.named_parameters():
        out["init." + n] = p.detach().float().numpy().copy()      # .float() of an fp32 tensor aliases the parameter
    lrs, norms = [], []
    for step in range(1, o["steps"] + 1):
        lr = get_lr_sched(step, opts)
        for g in opt.param_groups:
            g["lr"] = lr
        for n, p in model.named_parameters():
            if not n.startswith("unused"):
                p.grad = optim_toy_grad(n, p.shape, step).to(p.dtype)
        norms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), o["grad_norm"])))
        opt.step()
        opt.zero_grad()
        lrs.append(lr)
        for n, p in model.named_parameters():
            out["step%d.%s" % (step, n)] = p.detach().float().numpy().copy()
    out.update(lr=np.array(lrs), grad_norm=np.array(norms))
    np.savez_compressed(os.path.join(OUT, "optim_reduced.npz"), **out)
    print("optim: lr", lrs, "norms", [round(x, 4) for x in norms])


BACKBONE_ARGS = ("txt_ids", "txt_lens", "traj_view_img_fts", "traj_obj_img_fts", "traj_loc_fts", "traj_nav_types",
                 "traj_step_lens", "traj_vp_view_lens", "traj_vp_obj_lens", "traj_vpids", "traj_cand_vpids", "gmap_lens",
                 "gmap_step_ids", "gmap_pos_fts", "gmap_pair_dists", "gmap_vpids", "vp_pos_fts", "grid_fts", "grid_map")


def gen_backbone():
    """GlocalTextPathCMT.forward(...) / forward_mlm(...) of the imported reference (pretrain_src/model/vilmodel.py:
    668-856) called POSITIONALLY on the sap / mlm batches of pretrain_reduced.npz -> pretrain_backbone_reduced.npz."""
    import collections
    torch.set_num_threads(1)
    model = R.build_ref_pretrain_model(seed=9).eval()
    out = {"versions": _versions(), "weight_seed": 9, "cfg": json.dumps(dict(R.PRETRAIN_REDUCED))}
    with torch.no_grad():
        b = collections.defaultdict(lambda: None, pretrain_batch("sap"))
        g, v, m = model.bert(*[b[k] for k in BACKBONE_ARGS], gridmap_pos_fts=b["gridmap_pos_fts"])
        out.update(sap_gmap_embeds=g.float().numpy(), sap_vp_embeds=v.float().numpy(), sap_gridmap_embeds=m.float().numpy())
        g2, _, _ = model.bert(*[b[k] for k in BACKBONE_ARGS], gridmap_pos_fts=b["gridmap_pos_fts"], return_gmap_embeds=False)
        assert g2 is None
        b = collections.defaultdict(lambda: None, pretrain_batch("mlm"))
        t = model.bert.forward_mlm(*[b[k] for k in BACKBONE_ARGS], b["gridmap_pos_fts"])
        out.update(mlm_txt_embeds=t.float().numpy())
    np.savez_compressed(os.path.join(OUT, "pretrain_backbone_reduced.npz"), **out)
    print("backbone:", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


CLIP_CASES = {"reduced": dict(width=128, layers=2, heads=2, n_images=6, seed=21),
              "full": dict(width=768, layers=12, heads=12, n_images=2, seed=22)}


def clip_det_state(model, seed):
    """Deterministic CLIP weights keyed by state_dict name: LayerNorm gains near 1, everything else small."""
    sd = {}
    for k, v in model.state_dict().items():
        x = R.det_tensor("clip." + k, tuple(v.shape), seed)
        if ".ln_" in k or k.startswith("visual.ln_"):
            rs = np.random.RandomState((zlib_crc(k) ^ seed) & 0x7FFFFFFF)
            x = torch.from_numpy((1.0 + 0.05 * rs.standard_normal(tuple(v.shape))).astype(np.float32)) if k.endswith("weight") \
                else torch.from_numpy((0.02 * rs.standard_normal(tuple(v.shape))).astype(np.float32))
        sd[k] = x.to(v.dtype)
    return sd


def zlib_crc(name):
    import zlib
    return zlib.crc32(name.encode())


def clip_images(case):
    c = CLIP_CASES[case]
    return torch.from_numpy(np.random.RandomState(c["seed"]).standard_normal((c["n_images"], 3, 224, 224)).astype(np.float32))


def gen_clip():
    """VLN_CE/vlnce_baselines/models/gridmap/clip.py: CLIP(224, 32, width, layers, heads)(images) -> (N, 50, width) tokens
    for a reduced tower and for the full ViT-B/32 shape -> tests/golden/clip_tokens.npz (pins gridmm_amd/clip_encoder.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_gridmap_clip", os.path.join(R.REF_ROOT, "VLN_CE", "vlnce_baselines", "models", "gridmap", "clip.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                              # the reference's module
    out = {"versions": _versions(), "cases": json.dumps(CLIP_CASES)}
    torch.set_num_threads(8)
    for case, c in CLIP_CASES.items():
        torch.manual_seed(0)
        model = mod.CLIP(input_resolution=224, patch_size=32, width=c["width"], layers=c["layers"], heads=c["heads"]).eval()
        model.load_state_dict(clip_det_state(model, c["seed"]))
        with torch.no_grad():
            tok = model(clip_images(case))
        out[case + "_tokens"] = tok.numpy()
        out[case + "_keys"] = json.dumps(list(model.state_dict().keys()))
        print("clip", case, tuple(tok.shape), "abs max %.3f" % float(tok.abs().max()))
    np.savez_compressed(os.path.join(OUT, "clip_tokens.npz"), **out)


POLICY_CE = dict(B=2, steps=3, L=12, views=12, seed=31, cand_lens=[[4, 3], [3, 5], [4, 4]], N=[300, 200])


def policy_ce_inputs():
    """Scripted 3-step episode pair for GridMap.forward(mode='navigation') (shared by the generator and the test)."""
    c = POLICY_CE
    rs = np.random.RandomState(c["seed"])
    B, V = c["B"], c["views"]
    inp = {"lang_feats": torch.from_numpy(rs.standard_normal((B, c["L"], 768)).astype(np.float32) * 0.5),
           "lang_masks": torch.from_numpy(np.arange(c["L"])[None] < np.array([[c["L"]], [c["L"] - 4]])),
           "start": [tuple(rs.uniform(-3, 3, 3)) for _ in range(B)], "steps": []}
    pos = [np.array(p) for p in inp["start"]]
    for t in range(c["steps"]):
        cl = c["cand_lens"][t]
        nav_types = np.zeros((B, V), np.int64)
        for b in range(B):
            nav_types[b, :cl[b] - 1] = 1
        st = dict(positions=[tuple(p) for p in pos], headings=[float(rs.uniform(0, 2 * np.pi)) for _ in range(B)],
                  cand_lens=list(cl), angles=[list(rs.uniform(-np.pi, np.pi, cl[b] - 1)) for b in range(B)],
                  distances=[list(rs.uniform(0.5, 3.0, cl[b] - 1)) for b in range(B)],
                  view_img_fts=torch.from_numpy(rs.standard_normal((B, V, 768)).astype(np.float32) * 0.5),
                  loc_fts=torch.from_numpy(rs.standard_normal((B, V, 7)).astype(np.float32)),
                  nav_types=torch.from_numpy(nav_types), view_lens=torch.full((B,), V, dtype=torch.long),
                  grid_fts=[torch.from_numpy((rs.standard_normal((n, 768)) * 0.35).astype(np.float16).astype(np.float32)) for n in c["N"]],
                  grid_map=[torch.from_numpy(rs.randint(-1, 196, size=n).astype(np.float64)) for n in c["N"]],
                  gridmap_pos_fts=torch.stack([torch.from_numpy(G.gridmap_pos_fts(np.float32(rs.uniform(2, 9)))) for _ in range(B)]))
        inp["steps"].append(st)
        pos = [p + rs.uniform(-1.5, 1.5, 3) * np.array([1, 0.05, 1]) for p in pos]
    return inp


def gen_policy_ce():
    """VLN_CE/.../Policy_ViewSelection_GridMap.py GridMap.forward(mode='navigation') (:500-625) of the imported
    reference -- bare object (no habitat __init__), reduced VLN-CE model inside, `.cuda()` neutralised -- driven over
    three steps with the episode state set from outside as ss_trainer_GridMap.py:236-254 does -> policy_ce_nav.npz."""
    torch.set_num_threads(1)
    model = R.build_ref_vlnce_model(seed=7, **REDUCED)        # first: transformers must be imported before torchvision is stubbed
    mod = R.import_vlnce_policy()
    mod.DATASET, mod.MAX_DIST, mod.MAX_STEP = "R2R", 25, 20
    inp = policy_ce_inputs()
    g = object.__new__(mod.GridMap)
    g.__dict__["vln_bert"] = model
    B = POLICY_CE["B"]
    g.traj_embeds, g.traj_map = [[] for _ in range(B)], [[] for _ in range(B)]
    g.start_positions = inp["start"]
    keep = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = {"versions": _versions(), "weight_seed": 7, "cfg": json.dumps(REDUCED)}
    try:
        with torch.no_grad():
            for t, st in enumerate(inp["steps"]):
                g.positions, g.headings, g.action_step = st["positions"], st["headings"], t + 1
                logits = mod.GridMap.forward(
                    g, mode="navigation", lang_feats=inp["lang_feats"], lang_masks=inp["lang_masks"], positions=st["positions"],
                    candidate_lengths=st["cand_lens"], batch_angles=st["angles"], batch_distances=st["distances"],
                    batch_view_img_fts=st["view_img_fts"], batch_loc_fts=st["loc_fts"], batch_nav_types=st["nav_types"],
                    batch_view_lens=st["view_lens"], batch_grid_fts=st["grid_fts"], batch_map_index=st["grid_map"],
                    batch_gridmap_pos_fts=st["gridmap_pos_fts"])
                out["logits_%d" % t] = logits.numpy()
                print("policy_ce step", t, tuple(logits.shape))
    finally:
        torch.Tensor.cuda = keep
    np.savez_compressed(os.path.join(OUT, "policy_ce_nav.npz"), **out)


if __name__ == "__main__":
    assert R.reference_available(), "needs /root/reference"
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["fill", "nav", "navobj", "full", "textpano", "rollout", "vlnce", "pretrain", "navvlnce", "panoobj", "pretrainobj", "pretrainnotie", "topo", "optim", "backbone", "clip", "policyce"]
    if "rollout" in which: gen_rollout()
    if "rolloutfull" in which: gen_rollout(full=True)
    if "topo" in which: gen_topo_map()
    if "optim" in which: gen_optim()
    if "backbone" in which: gen_backbone()
    if "clip" in which: gen_clip()
    if "policyce" in which: gen_policy_ce()
    if "vlnce" in which: gen_fill_gridmap_vlnce()
    if "fill" in which: gen_fill_gridmap()
    if "nav" in which: gen_nav_reduced(False)
    if "navobj" in which: gen_nav_reduced(True)
    if "textpano" in which: gen_text_pano()
    if "textpanofull" in which: gen_text_pano(full=True)
    if "full" in which: gen_nav_full()
    if "pretrain" in which: gen_pretrain()
    if "navvlnce" in which: gen_nav_vlnce()
    if "navvlncefull" in which: gen_nav_vlnce(full=True)
    if "panoobj" in which: gen_pano_obj()
    if "pretrainobj" in which: gen_pretrain(True)
    if "pretrainfull" in which: gen_pretrain(full=True)
    if "pretrainnotie" in which: gen_pretrain(notie=True)
