"""The navigation loop (GMapNavAgent.rollout, agent.py:268-451) on scripted synthetic episodes.

Golden: rollout_reduced.npz = the same loop driven with the REFERENCE model (oracle/gen_golden.py).
CPU: loop + oracle model reproduces it (pins loop + oracle);  GPU: loop + HIP model + device grid memory."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden, golden_state_dict
from oracle import gen_golden
from oracle.adapters import OracleVLNBert


def _check(agent, traj, fx, tol):
    assert len(agent.trace) == int(fx["n_steps"])
    worst = 0.0
    for st in agent.trace:
        t = st["t"]
        live = ~fx["t%d_ended" % t]
        assert np.array_equal(st["ended"], fx["t%d_ended" % t])
        assert np.array_equal(st["a_t"][live], fx["t%d_a" % t][live]), "chosen actions differ at step %d" % t
        for key, name in (("fused_logits", "fused"), ("local_logits", "local"), ("global_logits", "global"),
                          ("grid_logits", "grid")):
            a, b = st["nav_outs"][key].detach().float().cpu().numpy(), fx["t%d_%s" % (t, name)]
            assert a.shape == b.shape
            fin = np.isfinite(b)
            assert np.array_equal(np.isfinite(a), fin)
            if fin[live].any():
                worst = max(worst, float(np.abs(a[live][fin[live]] - b[live][fin[live]]).max()))
    assert worst <= tol, worst
    assert json.dumps([t["path"] for t in traj]) == str(fx["traj"])      # same trajectories incl. stop backtrack
    return worst


def test_rollout_with_oracle_model_matches_reference_driven_golden():
    fx = load_golden("rollout_reduced.npz")
    agent = gen_golden.make_rollout_agent(OracleVLNBert(golden_state_dict(fx)))
    traj = agent.rollout()
    _check(agent, traj, fx, 5e-5)
    assert all(len(t["path"]) >= 1 for t in traj)


def test_teacher_forcing_follows_gt_path_and_accumulates_ce_loss():
    fx = load_golden("rollout_reduced.npz")
    agent = gen_golden.make_rollout_agent(OracleVLNBert(golden_state_dict(fx)))
    agent.feedback = "teacher"
    traj = agent.rollout(train_ml=1.0)
    for t, item in zip(traj, agent.env.batch):
        flat = [vp for seg in t["path"] for vp in seg]
        # teacher forcing walks towards the goal: the goal is reached within max_action_len or the walk is cut
        assert flat[0] == item["path"][0]
    assert float(agent.loss) > 0 and np.isfinite(float(agent.loss))


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["rollout_reduced.npz", "rollout_full.npz"])
def test_rollout_on_hip_matches_reference_driven_golden(fixture):
    """(rollout_full.npz: the released model size inside the loop -- same episodes, the reference model's per-step logits,
    actions and trajectories)"""
    from gridmm_amd.grid_memory import GridMemoryBatch
    from gridmm_amd.synthetic import NATIVE
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    fx = load_golden(fixture)
    model = GlocalTextPathNavCMT(default_config(**json.loads(str(fx["cfg"])))).cuda().eval()
    model.load_state_dict(golden_state_dict(fx), strict=True)
    r = gen_golden.ROLLOUT
    mem = GridMemoryBatch(r["batch_size"], NATIVE, max_steps=r["max_action_len"] + 2, device="cuda")
    agent = gen_golden.make_rollout_agent(model, device="cuda", grid_memory=mem)
    traj = agent.rollout()
    worst = _check(agent, traj, fx, 3e-4)     # north star: 1e-3
    print("max logit err over the rollout:", worst)


@pytest.mark.gpu
def test_dagger_training_iterations_on_hip():
    """Seq2SeqAgent.train (agent_base.py:164-211): teacher rollout (ml_weight) + sampled rollout -> one backward
    through language / panorama / navigation of every step -> clip 40 -> AdamW.  Checks: finite decreasing
    imitation loss on a fixed mini-batch, every parameter that the reference would train received a gradient,
    and the evaluation path afterwards sees the updated weights."""
    from gridmm_amd.agent import GMapNavAgent, default_args
    from gridmm_amd.grid_memory import GridMemoryBatch
    from gridmm_amd.sim_env import SyntheticNavEnv
    from gridmm_amd.synthetic import NATIVE
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    torch.manual_seed(0)
    np.random.seed(0)
    cfg = default_config(num_l_layers=2, num_pano_layers=1, num_x_layers=2, intermediate_size=256, vocab_size=2000,
                         hidden_dropout_prob=0.0)
    model = GlocalTextPathNavCMT(cfg).cuda()
    B = 4
    mem = GridMemoryBatch(B, NATIVE, max_steps=9, device="cuda")
    env = SyntheticNavEnv(B, mem, n_scans=2, n_episodes=B, seed=3)       # one fixed mini-batch, revisited
    args = default_args(max_action_len=7, train_alg="imitation", lr=2e-4)
    agent = GMapNavAgent(args, env, model, device="cuda")
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    losses = agent.train(6)
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses
    grads = {k: p.grad is not None for k, p in model.named_parameters()}
    gradless = sorted(k for k, g in grads.items() if not g)
    # no gradient, as in the reference (hence its DDP find_unused_parameters=True): sprel_linear is dead code on this
    # path (vilmodel.py:577-590) and grid_logits do not enter the fine-tune loss (agent.py:340-347 uses fused_logits)
    assert all(k.startswith(("global_encoder.sprel_linear", "grid_sap_head")) for k in gradless), gradless
    changed = sum(int(not torch.equal(before[k], v.detach())) for k, v in model.named_parameters())
    assert changed >= len(before) - len(gradless)
    # DAgger: two rollouts per iteration share one backward (the first rollout's slab must survive the reset)
    agent.args.train_alg, agent.args.ml_weight = "dagger", 0.2
    l2 = agent.train(2)
    assert all(np.isfinite(l2))
    res = agent.test()
    assert len(res) == B and all(len(r["trajectory"]) >= 1 for r in res)


@pytest.mark.gpu
def test_training_loop_without_read_backs_equals_the_synchronous_loop(monkeypatch):
    """Round 6: a teacher-forced training rollout issues no device read-back (stop probabilities stay on the device, the
    teacher's inputs come from the collator's host arrays, every upload goes through pinned rings), and train() fetches the
    losses once at the end -- the host runs an iteration ahead of the device.  Same arithmetic in the same order: losses, IL
    logs and parameters must be BIT-identical to the loop that reads back every step and every iteration
    (GRIDMM_TRAIN_SYNC=1 + traced rollouts), and a rollout called directly returns the same trajectories either way
    (the deferred stop-score bookkeeping, agent._resolve_stops)."""
    from gridmm_amd.agent import GMapNavAgent, default_args
    from gridmm_amd.grid_memory import GridMemoryBatch
    from gridmm_amd.sim_env import SyntheticNavEnv
    from gridmm_amd.synthetic import NATIVE
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config

    def make():
        torch.manual_seed(0)
        np.random.seed(0)
        cfg = default_config(num_l_layers=2, num_pano_layers=1, num_x_layers=2, intermediate_size=256, vocab_size=2000,
                             hidden_dropout_prob=0.0)
        model = GlocalTextPathNavCMT(cfg).cuda()
        mem = GridMemoryBatch(4, NATIVE, max_steps=9, device="cuda")
        env = SyntheticNavEnv(4, mem, n_scans=2, n_episodes=12, seed=3)
        return GMapNavAgent(default_args(max_action_len=7, train_alg="imitation", lr=2e-4), env, model, device="cuda"), model

    monkeypatch.setenv("GRIDMM_TRAIN_SYNC", "1")
    a, ma = make()
    la = a.train(5)
    monkeypatch.setenv("GRIDMM_TRAIN_SYNC", "0")
    b, mb = make()
    lb = b.train(5)
    assert la == lb and len(la) == 5 and all(np.isfinite(la)), (la, lb)
    assert a.logs["IL_loss"] == b.logs["IL_loss"] and len(b.logs["IL_loss"]) == 5
    for (k, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.equal(p, q), k
    # a direct teacher-forced rollout: deferred (one read-back at the end) vs per-step bookkeeping (a traced rollout)
    a.feedback = b.feedback = "teacher"
    a.env.reset_epoch(); b.env.reset_epoch()
    a.loss = b.loss = 0
    a.trace = []                                     # traced rollouts read the stop probabilities every step
    torch.manual_seed(5)                             # (the environment feature dropout of VLNBert.train draws from the global RNG)
    ta = a.rollout(train_ml=1.0)
    torch.manual_seed(5)
    tb = b.rollout(train_ml=1.0)
    assert ta == tb and any(len(t["path"]) > 1 for t in tb)
    assert float(a.loss) == float(b.loss)


class _StubMem:
    slab = True

    def reset(self):
        pass

    def step(self, *a):
        pass


class _StubStore:
    def append(self, mem, keys):
        return None, [(0.0, 0.0)] * len(keys)


class _StubModel:
    """Random embeddings / logits of the right shapes from a seeded generator: both collation paths see the same
    stream as long as they call the model with the same shapes in the same order."""

    def __init__(self, H=64):
        self.H, self.g = H, torch.Generator().manual_seed(0)

    def __call__(self, mode, b):
        if mode == "language":
            return torch.randn(*b["txt_ids"].shape, self.H, generator=self.g)
        if mode == "panorama":
            B, V = b["view_img_fts"].shape[:2]
            return torch.randn(B, V, self.H, generator=self.g), torch.arange(V)[None] < b["view_lens"][:, None]
        gm = b["gmap_masks"] & ~b["gmap_visited_masks"]
        gl = torch.randn(gm.shape, generator=self.g).masked_fill(~gm, -float("inf"))
        ll = torch.randn(b["vp_nav_masks"].shape, generator=self.g).masked_fill(~b["vp_nav_masks"], -float("inf"))
        return {"global_logits": gl, "local_logits": ll, "fused_logits": gl, "grid_logits": gl}


@pytest.mark.parametrize("over", [{}, {"enc_full_graph": False}, {"act_visited_nodes": True}])
def test_batched_collation_equals_the_per_episode_restatement(over):
    """collate.NavCollator (what the loop runs) against agent.py's line-by-line restatement of the reference's
    _panorama_feature_variable / _nav_gmap_variable / _nav_vp_variable + TopoMap node embeddings: every key of
    pano_inputs and nav_inputs at every step of a 10-step rollout -- integer / bool / name entries identical, floats
    within 1 ulp of 1.0 (the running mean is a multiplication by 1/count instead of a division)."""
    from gridmm_amd import synthetic as S
    from gridmm_amd.agent import GMapNavAgent, default_args
    from gridmm_amd.sim_env import SyntheticNavEnv

    def run(fast):
        env = SyntheticNavEnv(8, _StubMem(), n_scans=2, n_episodes=16, seed=3, geom=S.NATIVE, vocab=3000)
        env.device_store = _StubStore()
        ag = GMapNavAgent(default_args(max_action_len=10, **over), env, _StubModel(), device="cpu")
        ag.fast_collate, ag.trace = fast, []
        with torch.no_grad():
            traj = ag.rollout()
        return ag.trace, traj

    (a, ta), (b, tb) = run(False), run(True)
    assert len(a) == len(b) >= 5 and ta == tb
    for x, y in zip(a, b):
        for part in ("pano_inputs", "nav_inputs"):
            assert set(x[part]) == set(y[part]) - {"fusion_maps", "gmap_visited_masks_host"}
            if part == "nav_inputs":     # the host mirror the batched collator adds (teacher actions without a read-back)
                assert np.array_equal(np.asarray(y[part]["gmap_visited_masks_host"]), y[part]["gmap_visited_masks"].numpy())
            for k, v in x[part].items():
                w = y[part][k]
                if torch.is_tensor(v):
                    assert v.shape == w.shape and v.dtype == w.dtype, k
                    if v.dtype.is_floating_point:
                        assert torch.allclose(v, w, atol=2e-7, rtol=2e-7), (k, float((v - w).abs().max()))
                    else:
                        assert torch.equal(v, w), k
                elif k != "grid_memory":
                    assert v == w, k
        # the collator's integer maps of the logit fusion == the model's own host loops over the vpid lists
        from gridmm_amd.vilmodel import GlocalTextPathNavCMT
        n = x["nav_inputs"]
        want = GlocalTextPathNavCMT._fusion_index_maps(n["gmap_vpids"], n["gmap_visited_masks"], n["vp_cand_vpids"],
                                                       n["gmap_masks"].shape[1], n["vp_masks"].shape[1])
        got = y["nav_inputs"]["fusion_maps"]
        assert torch.equal(want[0], got[0]) and torch.equal(want[1], got[1])


def test_interleaved_rollouts_equal_sequential_rollouts():
    """GMapNavAgent.interleaved_rollouts: several rollouts advanced alternately at their per-step yield points give exactly
    the trajectories of running them one after the other (per agent, the order of calls does not change)."""
    from gridmm_amd import synthetic as S
    from gridmm_amd.agent import GMapNavAgent, default_args
    from gridmm_amd.sim_env import SyntheticNavEnv

    def agents():
        out = []
        for k in range(3):
            env = SyntheticNavEnv(4, _StubMem(), n_scans=2, n_episodes=8, seed=3 + k, geom=S.NATIVE, vocab=3000)
            env.device_store = _StubStore()
            out.append(GMapNavAgent(default_args(max_action_len=6 + 2 * k), env, _StubModel(), device="cpu"))
        return out
    with torch.no_grad():
        want = [a.rollout() for a in agents()]
        got = GMapNavAgent.interleaved_rollouts(agents())
    assert got == want and all(len(t) == 4 for t in got)


def test_native_navigation_collation_equals_the_numpy_form():
    """collate.NavCollator._navigation_batched (gridmm_collate_nav_plan / _fill, csrc/hostutil.hip) against its NumPy
    restatement on every step of a rollout: integer / bool / name entries identical, floats to 2e-7 (libm vs NumPy's own
    float32 sin / cos may differ in the last bit)."""
    from gridmm_amd import synthetic as S
    from gridmm_amd.agent import GMapNavAgent, default_args
    from gridmm_amd.sim_env import SyntheticNavEnv

    def run(native, **over):
        env = SyntheticNavEnv(8, _StubMem(), n_scans=2, n_episodes=16, seed=5, geom=S.NATIVE, vocab=3000)
        env.device_store = _StubStore()
        ag = GMapNavAgent(default_args(max_action_len=12, **over), env, _StubModel(), device="cpu")
        ag.collator.native, ag.trace = native, []
        with torch.no_grad():
            traj = ag.rollout()
        return ag.trace, traj
    for over in ({}, {"enc_full_graph": False}, {"act_visited_nodes": True}):
        (a, ta), (b, tb) = run(False, **over), run(True, **over)
        assert len(a) == len(b) >= 5 and ta == tb
        for x, y in zip(a, b):
            assert set(x["nav_inputs"]) == set(y["nav_inputs"])
            for k, v in x["nav_inputs"].items():
                w = y["nav_inputs"][k]
                if k == "fusion_maps":
                    assert torch.equal(v[0], w[0]) and torch.equal(v[1], w[1])
                elif torch.is_tensor(v):
                    assert v.shape == w.shape and v.dtype == w.dtype, k
                    if v.dtype.is_floating_point:
                        assert torch.allclose(v, w, atol=2e-7, rtol=2e-7), (k, float((v - w).abs().max()))
                    else:
                        assert torch.equal(v, w), k
                elif isinstance(v, np.ndarray):
                    assert np.array_equal(v, w), k
                elif k != "grid_memory":
                    assert v == w, k


@pytest.mark.gpu
def test_rollout_with_graph_replay_equals_the_eager_rollout():
    """GMapNavAgent.enable_graph_replay(): 'panorama' + the shape-dependent half of 'navigation' from hipGraphs keyed by
    shape, node / view axes padded to buckets, varlen cell buckets from the grid memory's tracked count -- same actions
    and trajectories as the eager rollout on the reference's shapes, logits of the real rows within 2e-5."""
    from gridmm_amd import synthetic as S
    from gridmm_amd.agent import GMapNavAgent, default_args
    from gridmm_amd.grid_memory import GridMemoryBatch
    from gridmm_amd.sim_env import SyntheticNavEnv
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    dev, B, T = torch.device("cuda"), 6, 8
    torch.manual_seed(0)
    cfg = default_config(num_l_layers=2, num_pano_layers=1, num_x_layers=2, intermediate_size=512, vocab_size=3000)
    model = GlocalTextPathNavCMT(cfg).eval().to(dev)
    model.varlen_buckets = GlocalTextPathNavCMT.DEFAULT_BUCKETS

    def run(graphs):
        mem = GridMemoryBatch(B, S.NATIVE, max_steps=T + 2, device=dev)
        env = SyntheticNavEnv(B, mem, n_scans=2, n_episodes=2 * B, seed=5, geom=S.NATIVE, vocab=3000)
        env.build_device_store(dev)
        ag = GMapNavAgent(default_args(max_action_len=T), env, model, device=dev)
        ag.feedback = "argmax"
        ag._set_mode(False)
        if graphs:
            ag.enable_graph_replay()
        ag.trace = []
        with torch.no_grad():
            traj = [ag.rollout(), ag.rollout()]          # second mini-batch: other instruction lengths, graphs reused
        return ag, traj

    (ea, te), (ga, tg) = run(False), run(True)
    assert te == tg and len(ea.trace) == len(ga.trace)
    for x, y in zip(ea.trace, ga.trace):
        assert np.array_equal(x["a_t"], y["a_t"])
        for k in ("fused_logits", "global_logits", "local_logits", "grid_logits"):
            a, b = x["nav_outs"][k], y["nav_outs"][k]
            b = b[:, :a.shape[1]]
            f = torch.isfinite(a)
            assert torch.equal(f, torch.isfinite(b)), k
            assert (a[f] - b[f]).abs().max() < 2e-5, (k, float((a[f] - b[f]).abs().max()))
        assert not torch.isfinite(y["nav_outs"]["fused_logits"][:, x["nav_outs"]["fused_logits"].shape[1]:]).any()
    g = ga._graphs[1]
    assert g.replays == len(ga.trace) and g.captures < g.replays
    lg = ga._graphs[2]                                    # forward('language') once per rollout, from a graph per (B, L)
    assert lg.replays == 2 and 1 <= lg.captures <= 2
    # D = 768 memory: the relevance of earlier observations is kept across the steps of an episode (cleared once per
    # rollout: a new instruction tensor), in the eager run and under the graphs alike
    for a in (ea, ga):
        assert a.env.grid_memory._rel is not None and a.env.grid_memory._rel["clears"] == 2
    # an in-place weight update (an optimizer step between two evaluations): the captured graphs hold the packed weight
    # planes of their capture, so the next rollout must drop and re-capture them
    with torch.no_grad():
        model.text_proj.weight.mul_(1.25)
        model.global_sap_head.net[0].weight.mul_(0.9)
    caps, env, graphs = g.captures, ga.env, ga._graphs
    ix0 = env.ix
    ga.trace = []
    with torch.no_grad():
        t1 = ga.rollout()
    tr1, ga.trace, ga._graphs, env.ix = ga.trace, [], None, ix0          # the same mini-batch again, eager launches
    with torch.no_grad():
        t2 = ga.rollout()
    assert t1 == t2 and g.captures > caps and len(tr1) == len(ga.trace)
    for x, y in zip(tr1, ga.trace):
        for k in ("fused_logits", "grid_logits"):
            f = torch.isfinite(y["nav_outs"][k])
            assert torch.equal(f, torch.isfinite(x["nav_outs"][k]))
            assert (x["nav_outs"][k][f] - y["nav_outs"][k][f]).abs().max() < 2e-5, k


@pytest.mark.gpu
def test_device_store_environment_equals_host_assembled_observations():
    """SyntheticNavEnv with the observations resident in HBM (feature_store.DeviceStore: device-side gather into the
    grid memory's next slot) against the same environment assembling every observation on the host and uploading it: the
    same grid memory bit for bit (slab, cell ids, sort, position features) and the same logits at every step."""
    from gridmm_amd import synthetic as S
    from gridmm_amd.agent import GMapNavAgent, default_args
    from gridmm_amd.grid_memory import GridMemoryBatch
    from gridmm_amd.sim_env import SyntheticNavEnv
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    dev, B, T = torch.device("cuda"), 5, 6
    torch.manual_seed(1)
    cfg = default_config(num_l_layers=1, num_pano_layers=1, num_x_layers=2, intermediate_size=256, vocab_size=3000)
    model = GlocalTextPathNavCMT(cfg).eval().to(dev)
    runs = []
    for dev_store in (False, True):
        mem = GridMemoryBatch(B, S.NATIVE, max_steps=T + 2, device=dev)
        env = SyntheticNavEnv(B, mem, n_scans=2, n_episodes=2 * B, seed=9, geom=S.NATIVE, vocab=3000)
        if dev_store:
            env.build_device_store(dev)
        ag = GMapNavAgent(default_args(max_action_len=T), env, model, device=dev)
        ag.feedback, ag.trace = "argmax", []
        ag._set_mode(False)
        with torch.no_grad():
            traj = ag.rollout()
        runs.append((ag.trace, traj, mem))
    (a, ta, ma), (b, tb, mb) = runs
    assert ta == tb and len(a) == len(b) >= 3
    for k in ("slab", "cell_id", "perm", "cell_start", "pos_fts", "n_pts"):
        assert torch.equal(getattr(ma, k), getattr(mb, k)), k
    for x, y in zip(a, b):
        for k in ("fused_logits", "global_logits", "local_logits", "grid_logits"):
            assert torch.equal(x["nav_outs"][k], y["nav_outs"][k]), (x["t"], k)


def test_panorama_cache_is_bounded_lru_and_equals_the_uncached_collation():
    """the device tables of collated panorama blocks are capped (pano_cache_slots) and recycle their least
    recently used slots; whatever the eviction history, a step's gathered batch equals the per-step assembly."""
    import numpy as np
    import torch
    from types import SimpleNamespace
    from gridmm_amd.collate import NavCollator
    fs, A, B = 16, 4, 2
    args = SimpleNamespace(image_feat_size=fs, angle_feat_size=A)
    rs = np.random.RandomState(0)
    world = {}

    def ob(vp, view):
        key = (vp, view)
        if key not in world:
            nc = int(rs.randint(0, 5))
            world[key] = dict(scan="s", viewpoint="v%d" % vp, viewIndex=view,
                              feature=rs.randn(36, fs + A).astype(np.float32),
                              candidate=[dict(pointId=int(p), viewpointId="c%d_%d" % (vp, p),
                                              feature=rs.randn(fs + A).astype(np.float32))
                                         for p in rs.choice(36, nc, replace=False)])
        return world[key]

    cached, plain = NavCollator(args, "cpu"), NavCollator(args, "cpu")
    cached.pano_cache_slots, plain.pano_cache = 5, False
    seq = [(0, 0), (1, 3), (2, 5), (0, 0), (3, 1), (4, 2), (5, 7), (1, 3), (6, 0), (0, 0), (7, 1), (2, 5)]
    for t in range(len(seq) - 1):
        obs = [ob(*seq[t]), ob(*seq[t + 1])]
        got, want = cached.panorama(obs), plain.panorama(obs)
        for k in ("view_img_fts", "loc_fts", "nav_types", "view_lens"):
            g, w = got[k], want[k]
            n = min(g.shape[1], w.shape[1]) if g.dim() > 1 else None
            if n is None:
                assert torch.equal(g, w), (t, k)
            else:
                assert torch.equal(g[:, :n], w[:, :n]), (t, k)
                assert not g[:, n:].any() and not w[:, n:].any()
        assert got["cand_vpids"] == want["cand_vpids"]
        P = cached._pano
        assert P["cap"] <= max(5, 2 * B) and len(P["slot"]) <= P["cap"] and len(P["key_of"]) <= P["cap"]
    assert len(world) > 5                       # more distinct states than slots: evictions happened
    cached.clear_panorama_cache()
    assert cached._pano is None
