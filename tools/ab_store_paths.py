"""A/B of the two observation paths of SyntheticNavEnv (host-assembled vs DeviceStore): grid memory contents and
step-0 logits in inference, then training losses."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gridmm_amd.agent import GMapNavAgent, default_args
from gridmm_amd.grid_memory import GridMemoryBatch
from gridmm_amd.sim_env import SyntheticNavEnv
from gridmm_amd import synthetic
from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config


def make(dev_store):
    geom = synthetic.NATIVE
    torch.manual_seed(0); np.random.seed(0)
    cfg = default_config(num_l_layers=2, num_pano_layers=1, num_x_layers=2, intermediate_size=512, vocab_size=30000)
    model = GlocalTextPathNavCMT(cfg).cuda()
    mem = GridMemoryBatch(8, geom, max_steps=9, device="cuda")
    env = SyntheticNavEnv(8, mem, n_scans=2, n_episodes=16, seed=3, geom=geom, vocab=30000)
    if dev_store:
        env.build_device_store("cuda")
    args = default_args(max_action_len=7, train_alg="imitation", lr=1e-5, feat_dropout=0.0, dropout=0.0)
    return GMapNavAgent(args, env, model, device="cuda"), mem, model


res = []
for dev_store in (False, True):
    agent, mem, model = make(dev_store)
    model.eval(); agent._set_mode(False); agent.feedback = "teacher"; agent.trace = []
    with torch.no_grad():
        agent.rollout()
    res.append((agent.trace, mem.slab.clone(), mem.cell_id.clone(), mem.perm.clone(), mem.pos_fts.clone(), mem.n_pts.clone()))
(a, sa, ca, pa, fa, na), (b, sb, cb, pb, fb, nb) = res
print("n_pts equal", torch.equal(na, nb), "slab equal", torch.equal(sa, sb), "cell ids equal", torch.equal(ca, cb), "perm equal",
      torch.equal(pa, pb), "pos_fts max diff", float((fa - fb).abs().max()))
for x, y in zip(a, b):
    d = {k: float((x["nav_outs"][k][torch.isfinite(x["nav_outs"][k])] - y["nav_outs"][k][torch.isfinite(y["nav_outs"][k])]).abs().max())
         for k in ("fused_logits", "grid_logits")}
    ins = {k: float((x["nav_inputs"][k].float() - y["nav_inputs"][k].float()).abs().max()) for k in ("gmap_img_embeds", "vp_img_embeds", "gmap_pos_fts", "vp_pos_fts")}
    print("t", x["t"], d, ins)
