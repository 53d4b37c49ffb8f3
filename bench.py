#!/usr/bin/env python
"""bench.py -- nav steps/s of the GridMM grid-memory hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic episodes, inputs resident in HBM:
    fill_gridmap (project the new 36x196 observation, re-bin the memory, build per-cell lists)
  + GlocalTextPathNavCMT.forward('navigation')  (aggregation, grid/cross-modal encoders, logit fusion)
Workload = BASELINE.json configs[1]: B=32 episodes per GPU, slab 36 views x 196 patches x 512-D (N=7056
points, memory depth t=1), L=80 instruction tokens, G=20 map nodes, 36 views + stop, full-size model
(161 M-parameter architecture with text_proj/grid_proj at D_in=512), random-init weights, synthetic data.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-execs itself under
                                                            torch.distributed.run, one rank per device)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Episodes are independent: ranks shard the episode batch, no collective on the step path ("weak" scaling,
B per GPU fixed).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch


def _dist_forced():
    """GRIDMM_DIST_FORCE as an integer switch (gridmm_amd.dist.dist_forced; restated here: this runs before the package import)."""
    try:
        return int(os.environ.get("GRIDMM_DIST_FORCE", "0").strip() or "0") != 0
    except ValueError:
        return False

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "nav steps/sec (whole node), R2R batch=32, 36\u00d7196\u00d7512 grid, 1/2/4/8 MI355X"   # byte-identical to BASELINE.json's "metric"
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="episodes per GPU")
    ap.add_argument("--shape", default="baseline", choices=["baseline", "native"])
    ap.add_argument("--mem-steps", type=int, default=1, help="observations in each episode's memory (t)")
    ap.add_argument("--eager", action="store_true", help="launch kernels eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-torch-gpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-depth-legs", action="store_true", help="skip the extra t = 5 / t = 15 timings")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the pre-training step timing (train_samples_per_s)")
    ap.add_argument("--train-leg-only", action="store_true", help="(internal) run the training leg and print its JSON")
    ap.add_argument("--no-producer-leg", action="store_true", help="skip the VLN-CE step with the CLIP tower in the timed region")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-relevance-cache", action="store_true",
                    help="two-pass aggregation shapes (--shape native: D = 768): recompute the relevance of every point at every "
                         "step like the reference instead of keeping it next to the device-resident slab")
    return ap.parse_args()


def build_workload(args, dev, mem_steps=None, device_feats=False, depth_mode="uniform", buckets=None, batch_size=None,
                   with_obj=False, n_obj=0, instruction_cache=False):
    """mem_steps overrides args.mem_steps (the extra t = 5 / 15 legs); device_feats fills the slab with N(0,1) drawn on
    the GPU instead of the host-generated features (no oracle leg runs on those workloads).  depth_mode / buckets: the
    sparse-map leg (synthetic.make_observations "ring", graph.GraphedNavStep buckets); batch_size / with_obj / n_obj:
    configs 1 and 4 (B = 1 latency, REVERIE object tokens)."""
    from gridmm_amd import synthetic as S
    from gridmm_amd.grid_memory import GridMemoryBatch
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config

    geom = S.BASELINE if args.shape == "baseline" else S.NATIVE
    torch.manual_seed(0)
    model = GlocalTextPathNavCMT(default_config(grid_feat_size=geom.feat_dim, obj_feat_size=768 if with_obj else 0)).eval()
    # BERT-style random init leaves LayerNorm at (1, 0); fine for timing
    model.to(dev)
    rs = np.random.RandomState(int(os.environ.get("RANK", "0")))
    B, t = (batch_size or args.batch), (args.mem_steps if mem_steps is None else mem_steps)
    host_batch = S.make_nav_batch(rs, B, L=80, G=20, n_visited=6, V1=37 + n_obj, n_cand=4, min_len=30, with_obj=with_obj,
                                  **({"n_obj": n_obj} if with_obj else {}))
    batch = S.batch_to(host_batch, dev)
    # what a caller has on the host each step for the fused-logit index maps (vilmodel.py:881-899)
    fusion_src = (host_batch["gmap_vpids"], host_batch["gmap_visited_masks"].numpy(), host_batch["vp_cand_vpids"])
    mem = GridMemoryBatch(B, geom, max_steps=t, device=dev)
    # two-pass aggregation shapes only (D = 768): the relevance of the points of earlier steps stays next to the slab; inside
    # the captured step too (every replay restores the same history prefix and appends the same observation)
    mem.relevance_cache_enabled = mem.relevance_cache_in_graphs = not getattr(args, "no_relevance_cache", False)
    okw = dict(depth_mode=depth_mode, inner_frac=0.004) if depth_mode != "uniform" else {}
    eps = [S.make_observations(rs, geom, t, with_feats=not device_feats, **okw) for _ in range(B)]
    n_new = geom.pts_per_obs
    depth = [torch.from_numpy(np.stack([e[k]["depth"].reshape(-1) for e in eps])).to(dev) for k in range(t)]
    # tokens are written into the slab once, before timing (zero-copy append: producer-owned slot)
    for k in range(t):
        if device_feats:
            mem.slab[:, k * n_new:(k + 1) * n_new].copy_(torch.randn(B, n_new, geom.feat_dim, device=dev))
        else:
            mem.slab[:, k * n_new:(k + 1) * n_new].copy_(torch.from_numpy(np.stack([e[k]["feats"] for e in eps])))
    poses = [[(e[k]["x"], e[k]["y"]) for e in eps] for k in range(t)]
    heads = [[e[k]["heading"] for e in eps] for k in range(t)]
    for k in range(t - 1):                      # history prefix (t-1 observations), built once
        mem.step(depth[k], None, poses[k], heads[k])
    restore = (mem.n_pts.clone(), mem.bbox.clone())
    n_host0 = mem.n_pts_host.copy()
    batch.update(grid_memory=mem, grid_fts=None, grid_map=None, gridmap_pos_fts=None)

    def eager_step():
        mem.n_pts.copy_(restore[0])
        mem.bbox.copy_(restore[1])
        mem.n_pts_host[:] = n_host0
        mem.step(depth[t - 1], None, poses[t - 1], heads[t - 1])   # project the new observation + re-bin all
        # the fused-logit index maps are rebuilt from the vpid lists on every call, as in the reference
        return model("navigation", dict(batch, fusion_maps=model.fusion_maps(
            dict(batch, gmap_visited_masks=fusion_src[1]), dev)))

    step = eager_step
    if not args.eager:
        from gridmm_amd.graph import GraphedNavStep
        eager_step()                            # packs the weights, fills the allocator
        g = GraphedNavStep(model, mem, batch, depth[t - 1], restore=restore, buckets=buckets, count_nodes=not buckets,
                           instruction_cache=instruction_cache)
        mem.n_pts_host[:] = n_host0 + n_new
        if buckets:
            g(poses[t - 1], heads[t - 1], fusion=fusion_src, check=True)     # settles the bucket prediction

        def step():   # host half (pose / heading floats, fused-logit index maps) + one graph replay
            return g(poses[t - 1], heads[t - 1], fusion=fusion_src, check=False)
        step.graph = g
    return model, batch, mem, eps, step, eager_step, geom


def time_steps(step, steps, warmup, dist):
    for _ in range(warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    from gridmm_amd.dist import max_over_ranks
    return max_over_ranks(dt)          # the slowest rank defines the step time (identity at N=1)


LOGIT_KEYS = ("global_logits", "local_logits", "grid_logits", "fused_logits")


def check_replay(step, eager_step):
    """What bench.py times is the hipGraph replay: compare its outputs with the same step launched eagerly (same
    kernels, same order: expected bit-identical) and fail loudly if they differ."""
    got = {k: v.clone() for k, v in step().items() if k in LOGIT_KEYS}
    torch.cuda.synchronize()
    want = eager_step()
    torch.cuda.synchronize()
    worst, bitwise = 0.0, True
    for k in LOGIT_KEYS:
        a, w = got[k], want[k]
        f = torch.isfinite(w)
        if not torch.equal(f, torch.isfinite(a)):
            raise SystemExit("bench.py: replayed %s has -inf in different places than the eager step" % k)
        bitwise &= bool(torch.equal(a[f], w[f]))
        if f.any():
            worst = max(worst, float((a[f] - w[f]).abs().max()))
    if worst > 1e-6:
        raise SystemExit("bench.py: replayed logits differ from the eager step by %.3g" % worst)
    return {"replay_vs_eager_max_abs": worst, "bit_identical": bitwise}


def extra_depth_leg(args, dev, dist, mem_steps, steps):
    """nav steps/s of this rank at memory depth t = mem_steps (slab filled on the device), same step otherwise."""
    model, batch, mem, eps, step, eager_step, geom = build_workload(args, dev, mem_steps=mem_steps, device_feats=True)
    dt = time_steps(step, steps, 2, dist)
    del model, batch, mem, step, eager_step
    torch.cuda.empty_cache()
    return dt / steps


def instruction_cache_leg(args, dev, steps, headline_ms):
    """The headline step with the instruction-side projections taken from the per-episode cache (computed once per episode
    at its first step, exactly as GMapNavAgent.rollout does through graph.NavigationGraphs): text_proj + fragments, the
    instruction's K / V of the grid / text layer, the 80 instruction rows of the local encoder's K / V GEMM.  Same logits
    bit for bit (tests/test_hip_instruction_cache.py).  A SECONDARY key: the headline recomputes them every step like the
    reference (map_nav_src/models/vilmodel.py:793, 841-853)."""
    model, batch, mem, eps, step, eager_step, geom = build_workload(args, dev, instruction_cache=True)
    out_cached = {k: v.clone() for k, v in step().items() if torch.is_tensor(v)}
    out_plain = eager_step()
    same = all(torch.equal(out_cached[k], out_plain[k]) for k in ("fused_logits", "global_logits", "local_logits"))
    dt = time_steps(step, steps, 2, None) / steps
    n_nodes = step.graph.n_nodes
    # the fill itself (once per episode): split + text_proj + fragments + 2 K/V GEMMs
    fill = _timed_loop(lambda i: model.instruction_cache(batch["txt_embeds"], batch["txt_masks"]), 10, None)
    res = {"value": args.batch / dt, "unit": "steps/s", "ms_per_step": 1e3 * dt, "graph_nodes": n_nodes,
           "fill_ms_per_episode": 1e3 * fill, "bit_identical_to_recompute": bool(same),
           "saved_ms_per_step": headline_ms - 1e3 * dt,
           "note": "per-episode constants computed once (as the rollout does); the headline value recomputes them every step"}
    del model, batch, mem, step, eager_step
    torch.cuda.empty_cache()
    return res


def larger_batch_leg(args, dev, steps):
    """The same step with several of the reference's batches resident at once (288 GB of HBM hold them easily): B = 64 and
    B = 128 episodes per GPU, t = 1.  Secondary keys: the metric is quoted at B = 32; the thin GEMMs of the local encoder
    (57 query rows per episode) fill the chip better with more episodes per launch."""
    res = {}
    for B in (64, 128):
        model, batch, mem, eps, step, eager_step, geom = build_workload(args, dev, device_feats=True, batch_size=B)
        dt = time_steps(step, steps, 2, None) / steps
        res["b%d" % B] = {"value": B / dt, "unit": "steps/s", "ms_per_step": 1e3 * dt, "batch": B}
        del model, batch, mem, step, eager_step
        torch.cuda.empty_cache()
    return res


def sparse_map_leg(args, dev, dist, steps):
    """Varlen map sequences (vilmodel.py:809-823 max_cell_num): the headline step on episodes whose depth occupies ~90-120
    of the 196 cells (synthetic 'ring' depth), once on the 196-row padded sequence and once with the bucketed back graphs."""
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT
    res = {}
    for name, buckets in (("padded_196", None), ("bucketed", GlocalTextPathNavCMT.DEFAULT_BUCKETS)):
        model, batch, mem, eps, step, eager_step, geom = build_workload(args, dev, device_feats=True, depth_mode="ring",
                                                                        buckets=buckets)
        dt = time_steps(step, steps, 2, dist) / steps
        res[name] = {"ms_per_step": 1e3 * dt, "value": args.batch / dt}
        if buckets:
            g = step.graph
            cmax = int(g.cmax[g.last_bucket].item())
            if cmax > g.last_bucket:
                raise SystemExit("bench.py: sparse-map leg ran on bucket %d with %d occupied cells" % (g.last_bucket, cmax))
            res[name].update(bucket=g.last_bucket, cmax=cmax, buckets=list(buckets))
            a = {k: v.clone() for k, v in step().items() if k in LOGIT_KEYS}
            torch.cuda.synchronize()
            b = eager_step()                       # the 196-row eager path on the same inputs
            worst = 0.0
            for k in LOGIT_KEYS:
                f = torch.isfinite(b[k])
                if not torch.equal(f, torch.isfinite(a[k])):
                    raise SystemExit("bench.py: bucketed %s masks differ from the padded path" % k)
                worst = max(worst, float((a[k][f] - b[k][f]).abs().max()))
            if worst > 1e-4:
                raise SystemExit("bench.py: bucketed logits differ from the padded path by %.3g" % worst)
            res[name]["max_abs_vs_padded"] = worst
        del model, batch, mem, step, eager_step
        torch.cuda.empty_cache()
    res["speedup"] = res["padded_196"]["ms_per_step"] / res["bucketed"]["ms_per_step"]
    res["workload"] = "the headline step on 'ring' depth (walls at ~3 m + 0.4 % nearer points): ~90-120 occupied cells"
    return res


def config_legs(args, dev, steps):
    """BASELINE.json configs[0] and configs[3] at their stated sizes, same step, full-size model:
    b1_latency: B = 1 episode (the val_unseen plumbing config): wall time of ONE step incl. its host half, a device
                synchronize after every step (latency, not throughput);
    reverie_b16: B = 16, 36 views + 21 object tokens (V1 = 58), obj_logits from og_head (map_nav_src/reverie/env.py:263-372,
                vilmodel.py:745-764, 903-907)."""
    res = {}
    model, batch, mem, eps, step, eager_step, geom = build_workload(args, dev, device_feats=True, batch_size=1)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    lat = []
    for _ in range(max(steps, 20)):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    lat.sort()
    res["b1_latency"] = {"latency_ms_median": 1e3 * lat[len(lat) // 2], "latency_ms_min": 1e3 * lat[0], "batch": 1,
                         "steps_per_s": 1.0 / lat[len(lat) // 2], "replay_check": check_replay(step, eager_step)}
    del model, batch, mem, step, eager_step
    torch.cuda.empty_cache()
    model, batch, mem, eps, step, eager_step, geom = build_workload(args, dev, device_feats=True, batch_size=16,
                                                                    with_obj=True, n_obj=21)
    dt = time_steps(step, steps, 2, None) / steps
    out = step()
    torch.cuda.synchronize()
    obj = out["obj_logits"]
    assert obj is not None and obj.shape == (16, 58) and bool(torch.isfinite(obj).any())
    res["reverie_b16"] = {"value": 16 / dt, "unit": "steps/s", "ms_per_step": 1e3 * dt, "batch": 16, "V1": 58, "object_tokens": 21,
                          "replay_check": check_replay(step, eager_step)}
    del model, batch, mem, step, eager_step
    torch.cuda.empty_cache()
    return res


def rollout_leg(args, dev, rollouts=4, profiler=None):
    """End to end: GMapNavAgent.rollout (map_nav_src/r2r/agent.py:268-451) over the synthetic environment, B = 32
    episodes x up to 15 steps at the BASELINE observation shape -- 'language' once, then per step 'panorama', TopoMap
    update, input collation, fill_gridmap, 'navigation', action selection, env step (argmax feedback, no_grad, varlen map
    sequences on).  value = episode-steps / wall second of whole rollouts; the second pass synchronises around every
    section to show where the wall time goes (host sections are Python)."""
    from gridmm_amd import synthetic as S
    from gridmm_amd.agent import GMapNavAgent, default_args
    from gridmm_amd.grid_memory import GridMemoryBatch
    from gridmm_amd.sim_env import SyntheticNavEnv
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    geom, B, T = S.BASELINE, args.batch, 15
    torch.manual_seed(0)
    model = GlocalTextPathNavCMT(default_config(grid_feat_size=geom.feat_dim)).eval().to(dev)
    model.varlen_buckets = GlocalTextPathNavCMT.DEFAULT_BUCKETS
    mem = GridMemoryBatch(B, geom, max_steps=T + 2, device=dev)
    env = SyntheticNavEnv(B, mem, n_scans=4, n_episodes=4 * B, seed=3, geom=geom, vocab=30000)
    env.build_device_store(dev)               # observations resident in HBM: an env step moves no feature bytes over PCIe
    agent = GMapNavAgent(default_args(max_action_len=T), env, model, device=dev)
    agent.feedback = "argmax"
    agent._set_mode(False)
    agent.enable_graph_replay()               # 'panorama' / 'navigation' from hipGraphs keyed by (bucketed) shape
    with torch.no_grad():
        for _ in range(4):                    # one epoch of the 4 mini-batches: fills the environment's feature memo, packs
            agent.rollout()                   # the weights, captures the graphs of the shapes these rollouts visit
        torch.cuda.synchronize()
        n0, t0 = agent.nav_steps, time.perf_counter()
        if profiler is not None:              # tools/bench_rollout.py --profile: the host side of the timed rollouts only
            profiler.enable()
        for _ in range(rollouts):
            agent.rollout()
        if profiler is not None:
            profiler.disable()
        torch.cuda.synchronize()
        dt, steps = time.perf_counter() - t0, agent.nav_steps - n0
        agent.timers = {}
        agent.rollout()                       # (the synchronised mode frees its temporaries at other moments than the timed mode:
        torch.cuda.synchronize()              #  its first rollout pays one-off device allocations, 60-80 ms -- not a section's cost)
        agent.timers = {}
        n1 = agent.nav_steps
        agent.rollout()
        torch.cuda.synchronize()
        prof_steps = agent.nav_steps - n1
    tot = sum(agent.timers.values())
    x2 = None
    try:
        # (B episodes as two half-batches in flight, rollout_interleaved(..., B=B // 2), measured in round 5: 8 250 against
        # 8 040 for the single batch on the same box -- the host work of a step does not halve with the batch)
        x2 = rollout_interleaved(args, dev, model, geom, T, rollouts)
    except Exception as e:          # a secondary key of a secondary key
        x2 = {"error": repr(e)[:200]}
    return {"value": B * steps / dt, "unit": "episode-steps/s", "ms_per_step": 1e3 * dt / steps, "batch": B,
            "two_interleaved_batches": x2,
            "steps_per_rollout": steps / rollouts, "max_action_len": T,
            "sections_ms_per_step": {k: 1e3 * v / prof_steps for k, v in sorted(agent.timers.items(), key=lambda kv: -kv[1])},
            "host_share": sum(v for k, v in agent.timers.items() if k.startswith("host") or k.startswith("env")) / tot,
            "workload": "GMapNavAgent.rollout, synthetic buildings (24 viewpoints, 36 views, 36x196x512 observations), "
                        "argmax actions, hipGraph replay per shape bucket (%d navigation graphs captured, %d replays), varlen "
                        "map sequences, batched host collation, observation store resident in HBM"
                        % (agent._graphs[1].captures, agent._graphs[1].replays)}


def finetune_leg(args, dev, iters=3):
    """Config 2 read as what it says, "R2R FINE-TUNE batch=32": one Seq2SeqAgent.train iteration (map_nav_src/r2r/
    agent_base.py:164-211, agent.py:268-451) -- zero_grad, a teacher-forced rollout of B = 32 episodes over the synthetic
    environment at the BASELINE observation shape ('language' once, then 'panorama' + fill_gridmap + 'navigation' per step on
    the DIFFERENTIABLE path, cross-entropy against the teacher action), ONE backward through every step of the rollout,
    clip_grad_norm 40, AdamW.  value = episodes / s; kernels_ms = summed device time of the iteration's kernels (HIP events
    around the whole iteration minus nothing: wall with a synchronize, the iteration is device-bound)."""
    from gridmm_amd import synthetic as S
    from gridmm_amd.agent import GMapNavAgent, default_args
    from gridmm_amd.grid_memory import GridMemoryBatch
    from gridmm_amd.sim_env import SyntheticNavEnv
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    geom, B, T = S.BASELINE, args.batch, 7
    torch.manual_seed(0)
    np.random.seed(0)
    model = GlocalTextPathNavCMT(default_config(grid_feat_size=geom.feat_dim)).to(dev)
    mem = GridMemoryBatch(B, geom, max_steps=T + 2, device=dev)
    env = SyntheticNavEnv(B, mem, n_scans=4, n_episodes=4 * B, seed=3, geom=geom, vocab=30000)
    env.build_device_store(dev)
    agent = GMapNavAgent(default_args(max_action_len=T, train_alg="imitation", lr=1e-5), env, model, device=dev)
    agent.train(8)                            # two passes over the episode list: feature memo, weight packs, allocator, and the
    #                                           pinned upload rings at the size a loop that runs ahead of the device needs
    torch.cuda.synchronize()
    n0 = agent.nav_steps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    losses = agent.train(iters)
    e1.record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    steps = (agent.nav_steps - n0) / iters
    del agent, env, mem, model
    torch.cuda.empty_cache()
    return {"value": B / dt, "unit": "episodes/s", "s_per_iteration": dt, "device_span_ms": e0.elapsed_time(e1) / iters,
            "nav_steps_per_iteration": steps, "episode_steps_per_s": B * steps / dt, "batch": B, "max_action_len": T,
            "finite_losses": bool(np.isfinite(losses).all()),
            "workload": "GMapNavAgent.train (imitation): teacher-forced rollout of %d episodes x <= %d steps on the "
                        "differentiable path, one backward through all steps, clip 40 + AdamW; 36x196x512 observations "
                        "resident in HBM, full-size model" % (B, T)}


def rollout_interleaved(args, dev, model, geom, T, rollouts, B=None, n_agents=2):
    """Evaluation throughput with SEVERAL mini-batches in flight (GMapNavAgent.interleaved_rollouts): the rollouts are advanced
    alternately at their per-step yield points, each on its own stream, so one batch's host collation runs under the
    others' 'navigation' kernels.  Same model, separate environments / grid memories / graph caches; every trajectory
    equals the one the batch produces alone (tests/test_agent_loop.py).  Default: two batches of B = 32; B = 16 x 2 is the
    reference's 32 episodes in flight, stepped as two half-batches."""
    from gridmm_amd.agent import GMapNavAgent, default_args
    from gridmm_amd.grid_memory import GridMemoryBatch
    from gridmm_amd.sim_env import SyntheticNavEnv
    B = B or args.batch
    agents = []
    for k in range(n_agents):
        mem = GridMemoryBatch(B, geom, max_steps=T + 2, device=dev)
        env = SyntheticNavEnv(B, mem, n_scans=4, n_episodes=4 * B, seed=3 + 7 * k, geom=geom, vocab=30000)
        env.build_device_store(dev)
        a = GMapNavAgent(default_args(max_action_len=T), env, model, device=dev)
        a.feedback = "argmax"
        a._set_mode(False)
        a.enable_graph_replay()
        agents.append(a)
    streams = [torch.cuda.Stream() for _ in agents]
    with torch.no_grad():
        for _ in range(4):
            GMapNavAgent.interleaved_rollouts(agents, streams)
        torch.cuda.synchronize()
        n0, t0 = sum(a.nav_steps for a in agents), time.perf_counter()
        for _ in range(rollouts):
            GMapNavAgent.interleaved_rollouts(agents, streams)
        torch.cuda.synchronize()
        dt, steps = time.perf_counter() - t0, sum(a.nav_steps for a in agents) - n0
    del agents
    torch.cuda.empty_cache()
    return {"value": B * steps / dt, "unit": "episode-steps/s", "ms_per_step_of_32": 1e3 * dt / (B * steps) * 32,
            "batches_in_flight": n_agents, "batch": B, "episodes_in_flight": n_agents * B}


def producer_leg(args, dev, steps=5):
    """Config 5's shape with the PRODUCER in the timed region: per step the CLIP ViT-B/32 tower encodes the
    12 view images of every episode (B x 12 x 3 x 224 x 224, already normalised and resident), writes the patch tokens
    into the grid memory's next slot, then fill_gridmap + forward('navigation') run as usual -- VLN-CE geometry
    (12 views x 49 patches x 768-D, habitat depth), full-size model + full-size tower, random init, t = 1."""
    from gridmm_amd import synthetic as S
    from gridmm_amd.clip_encoder import CLIP
    from gridmm_amd.grid_memory import GridMemoryBatch
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    geom, B = S.VLNCE_R2R, args.batch
    torch.manual_seed(1)
    model = GlocalTextPathNavCMT(default_config(grid_feat_size=geom.feat_dim)).eval().to(dev)
    clip = CLIP().eval().to(dev)
    rs = np.random.RandomState(5)
    host_batch = S.make_nav_batch(rs, B, L=80, G=20, n_visited=6, V1=37, n_cand=4, min_len=30)
    batch = S.batch_to(host_batch, dev)
    mem = GridMemoryBatch(B, geom, max_steps=1, device=dev)
    eps = [S.make_observations(rs, geom, 1, with_feats=False)[0] for _ in range(B)]
    depth = torch.from_numpy(np.stack([e["depth"].reshape(-1) for e in eps]).astype(np.float32)).to(dev)
    poses, heads = [(e["x"], e["y"]) for e in eps], [e["heading"] for e in eps]
    images = torch.randn(B * geom.n_views, 3, 224, 224, device=dev)
    batch.update(grid_memory=mem, grid_fts=None, grid_map=None, gridmap_pos_fts=None)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def step():
        mem.reset()
        ev[0].record()
        clip.encode_into(images, mem.next_slot(), n_views=geom.n_views)
        ev[1].record()
        mem.step(depth, None, poses, heads)
        out = model("navigation", dict(batch, fusion_maps=model.fusion_maps(
            dict(batch, gmap_visited_masks=host_batch["gmap_visited_masks"].numpy()), dev)))
        ev[2].record()
        return out
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enc = nav = 0.0
    for _ in range(steps):
        step()
        torch.cuda.synchronize()
        enc += ev[0].elapsed_time(ev[1])
        nav += ev[1].elapsed_time(ev[2])
    dt = (time.perf_counter() - t0) / steps
    flops = B * geom.n_views * (2.0 * 49 * 3072 * 768 + 12 * 50 * 2.0 * 768 * (2304 + 768 + 3072 + 3072) + 12 * 4.0 * 50 * 50 * 768)
    return {"value": B / dt, "unit": "steps/s", "ms_per_step": 1e3 * dt, "encoder_ms": enc / steps, "fill_nav_ms": nav / steps,
            "encoder_tflops_algorithmic": flops / (enc / steps * 1e-3) / 1e12, "launch": "eager",
            "workload": "B=%d episodes x 12 views: CLIP ViT-B/32 (224 px) -> slab, fill_gridmap (VLN-CE geometry) + "
                        "forward('navigation'), t=1, full-size model and tower, random init" % B}


def _init_dist(dev):
    """(dist module or None, rank, world): RCCL ("nccl") process group, or gloo when N ranks share one GPU (test hook)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1 and not _dist_forced():
        return None, 0, 1
    # (GRIDMM_DIST_FORCE=1 with one rank: the multi-rank leg end to end over the real backend -- the only way it can meet
    # RCCL on a one-GPU box, which refuses two ranks on one device; tests/test_hip_dist.py)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(_free_port()))
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    import torch.distributed as dist
    if os.environ.get("GRIDMM_BENCH_SHARE_GPU"):
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)
    return dist, rank, world


def _timed_loop(fn, n, dist):
    """Seconds per call of fn over n calls, bracketed by barrier + device synchronize, MAX over ranks."""
    from gridmm_amd.dist import max_over_ranks
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    return max_over_ranks(time.perf_counter() - t0) / n


def exchange_sweep(reducer, dev, dist, iters=5):
    """The gradient exchange alone (same bucket sizes as the training step), per algorithm and payload: ms per step's
    worth of gradients, max over ranks.  Evidence for SURVEY 8e's ring vs direct reduce-scatter / all-gather estimate."""
    res = {}
    share = bool(os.environ.get("GRIDMM_BENCH_SHARE_GPU"))
    flats = [torch.randn(b["padded"], device=dev) * 1e-3 for b in reducer.buckets]
    keep = (reducer.algo, reducer.payload)
    for algo in ("ring", "direct") if share else ("ring", "rsag", "direct"):
        for payload in ("fp32", "bf16"):
            reducer.algo, reducer.payload = algo, payload
            scratch = [dict() for _ in flats]

            def once(_):
                for f, k in zip(flats, scratch):
                    reducer._exchange(f, k)
            try:                             # (a collective this RCCL build rejects fails on every rank alike)
                once(0)
                res["%s_%s" % (algo, payload)] = 1e3 * _timed_loop(once, iters, dist)
            except Exception as e:
                res["%s_%s" % (algo, payload)] = "failed: " + repr(e)[:160]
    reducer.algo, reducer.payload = keep
    return res


def train_leg(args, dev, steps=None, emit=None):
    """One pre-training step (config 3's per-GPU shape) -- forward + backward + [gradient exchange] + gradient clip + fused
    AdamW of the full-size GlocalTextPathCMTPreTraining, B = 32 per rank, native grid memory of 3-5 observations, tasks
    cycling mlm / mrc / sap as the task-mixed loop does (pretrain_src/train_r2r.py:231-303).  Timed launched eagerly from
    Python and from hipGraphs (gridmm_amd/train_graph.py: same kernels, same updates; lr schedule, AdamW bias correction
    and dropout seeds advance per replay).  With WORLD_SIZE > 1 every rank trains its own batch and
    gridmm_amd.dist.GradientReducer exchanges the gradients (the DDP all-reduce of pretrain_src/utils/misc.py:52-65):
    eager = buckets launched from the backward hooks; graph = forward graph + one graph per backward segment, complete
    buckets launched on the side stream between segment launches, eager clip + AdamW.  Both are also timed WITHOUT the
    exchange (exposed_ms_* = the difference) and the graph step is timed once per exchange algorithm; value = whole-job
    samples/s (max-over-ranks time)."""
    from gridmm_amd.pretrain_cmt import GlocalTextPathCMTPreTraining
    from gridmm_amd.pretrain_loop import PreTrainer, default_opts
    from gridmm_amd.synthetic import batch_to, make_pretrain_batch
    from gridmm_amd.train_graph import GraphedTrainStep
    from gridmm_amd.vilmodel import default_config
    dist, rank, world = _init_dist(dev)
    multi = dist is not None                 # several ranks (or one rank with GRIDMM_DIST_FORCE: the same code over RCCL)
    share = bool(os.environ.get("GRIDMM_BENCH_SHARE_GPU"))
    steps = steps or int(os.environ.get("GRIDMM_BENCH_TRAIN_STEPS", "6"))
    cfg = default_config(use_lang2visn_attn=True, pretrain_tasks=["mlm", "mrc", "sap"], image_prob_size=1000, obj_prob_size=0)
    torch.manual_seed(0)
    model = GlocalTextPathCMTPreTraining(cfg).to(dev)
    # The first pass uses the ring all_reduce (the one collective every RCCL build runs) so that a number exists whatever
    # happens next; the captured step is then re-timed with the direct reduce-scatter / all-gather (all_to_all_single: every
    # xGMI link at once), its bf16 payload and reduce_scatter + all_gather, and the headline keys are those of the fastest
    # fp32 algorithm.  A collective that raises is recorded and ends the sweep.
    rkw = dict(algo=os.environ.get("GRIDMM_EXCHANGE_ALGO", "ring"), payload=os.environ.get("GRIDMM_EXCHANGE_PAYLOAD", "fp32"))
    tr = PreTrainer(model, default_opts(warmup_steps=100), reducer_kw=rkw)
    tasks = ("mlm", "mrc", "sap")
    batches = {t: batch_to(make_pretrain_batch(np.random.RandomState(i + 10 * rank), args.batch, t, max_steps=5, L=80, vocab=30000,
                                               image_prob_size=1000, n_pts=(588 * 3, 588 * 5)), dev)
               for i, t in enumerate(tasks)}
    fallback = None
    try:
        for t in tasks:                      # first sight of each task agrees on its used-set ...
            tr.train_step(batches[t], t)
    except Exception as e:                   # (a collective this RCCL build rejects fails on every rank alike)
        if not multi or tr.reducer.algo == "ring":
            raise
        fallback = "%s failed (%s): ring all_reduce instead" % (tr.reducer.algo, repr(e)[:200])
        tr.reducer.algo = "ring"
        tr.optimizer.zero_grad()
        for t in tasks:
            tr.train_step(batches[t], t)
    for t in tasks:                          # ... the second launches its buckets from inside backward
        tr.train_step(batches[t], t)

    def resync():
        """After a timing loop without the exchange the ranks have drifted apart: rank 0's weights and AdamW moments again."""
        if multi:
            from gridmm_amd.dist import broadcast_parameters
            broadcast_parameters(model.parameters())
            st = [v for p in model.parameters() for k, v in sorted(tr.optimizer.state.get(p, {}).items()) if torch.is_tensor(v)]
            broadcast_parameters(st)
    dt_eager = _timed_loop(lambda i: tr.train_step(batches[tasks[i % 3]], tasks[i % 3]), steps, dist)
    dt_noex = None
    if multi:
        tr.exchange = False                  # the same step without the exchange (ranks drift apart: timing only)
        dt_noex = _timed_loop(lambda i: tr.train_step(batches[tasks[i % 3]], tasks[i % 3]), steps, dist)
        tr.exchange = True
        resync()
    graphs = {t: GraphedTrainStep(tr, batches[t], t) for t in tasks}
    for t in tasks:
        graphs[t]()
    n = 4 * steps if not multi else 2 * steps
    last = {}

    def gstep(i):
        last["losses"], _ = graphs[tasks[i % 3]]()
    dt = _timed_loop(gstep, n, dist)
    seg_launches = list(getattr(graphs[tasks[0]], "launched_after_segment", []))
    finite = bool(torch.isfinite(last["losses"]).all())
    if not finite:
        raise SystemExit("bench.py: the captured training step produced non-finite losses")
    res = {"train_samples_per_s": world * args.batch / dt, "ms_per_step": 1e3 * dt, "batch": args.batch, "n_gpus": world,
           "global_batch": world * args.batch,
           "launch": ("hipGraph replay, one graph per task (train_graph.GraphedTrainStep)" if not multi else
                      "hipGraph replay: forward graph + %d backward-segment graphs per task, gradient buckets exchanged on a "
                      "side stream between segment launches, eager clip + AdamW" % (len(graphs[tasks[0]].graphs) - 1)),
           "graph": {"train_samples_per_s": world * args.batch / dt, "ms_per_step": 1e3 * dt},
           "eager": {"train_samples_per_s": world * args.batch / dt_eager, "ms_per_step": 1e3 * dt_eager},
           "best_samples_per_s": world * args.batch / min(dt, dt_eager),
           "workload": "pre-training step (mlm/mrc/sap cycling), full-size model, native 12x49x768 grid memory t=3..5, "
                       "fwd + bwd + clip + fused AdamW"}
    if multi:
        tr.exchange = False
        dt_graph_noex = _timed_loop(gstep, n, dist)
        tr.exchange = True
        resync()
        exposed = max(0.0, 1e3 * (dt_eager - dt_noex))
        exposed_g = max(0.0, 1e3 * (dt - dt_graph_noex))
        res["exchange"] = {"world": world, "algo": tr.reducer.algo, "payload": tr.reducer.payload,
                           "buckets": len(tr.reducer.buckets),
                           "gradient_mb": sum(b["numel"] for b in tr.reducer.buckets) * 4 / 1e6,
                           "exposed_ms_eager": exposed, "ms_per_step_without_exchange": 1e3 * dt_noex,
                           "exposed_ms_graph": exposed_g, "graph_ms_per_step_without_exchange": 1e3 * dt_graph_noex,
                           "buckets_launched_after_segment": seg_launches,
                           "reducer_stats": dict(tr.reducer.stats),
                           "backend": "gloo (ranks share one GPU: test hook; blocking collectives on a helper thread)" if share else "nccl (RCCL)"}
        if fallback:
            res["exchange"]["fallback"] = fallback
        if emit is not None and rank == 0:
            emit(res)                        # the step timings are out before anything else is tried
        # the captured step once per exchange algorithm (the exchange is launched eagerly: no re-capture)
        by_algo = {"%s_%s" % (tr.reducer.algo, tr.reducer.payload): 1e3 * dt}
        keep = (tr.reducer.algo, tr.reducer.payload)
        cands = [("ring", "fp32"), ("direct", "fp32")] + ([] if share else [("direct", "bf16"), ("rsag", "fp32")])
        for algo, payload in cands:
            key = "%s_%s" % (algo, payload)
            if key in by_algo or (fallback and algo != "ring"):
                continue
            tr.reducer.algo, tr.reducer.payload = algo, payload
            try:
                gstep(0)
                by_algo[key] = 1e3 * _timed_loop(gstep, max(1, n // 2), dist)
            except Exception as e:
                by_algo[key] = "failed: " + repr(e)[:160]
                break                        # (a failed collective may leave the communicator unusable: stop here)
        tr.reducer.algo, tr.reducer.payload = keep
        timed = {k: v for k, v in by_algo.items() if isinstance(v, float)}
        fastest = min(timed, key=timed.get)
        f32 = {k: v for k, v in timed.items() if k.endswith("fp32")}
        head = min(f32, key=f32.get)             # headline = the fastest exchange that keeps fp32 gradients on the wire
        res["exchange"].update(graph_ms_per_step_by_algo=by_algo, fastest_graph_step=fastest, headline_algo=head,
                               exposed_ms_graph_by_algo={k: max(0.0, v - 1e3 * dt_graph_noex) for k, v in timed.items()})
        res.update(train_samples_per_s=world * args.batch / (1e-3 * f32[head]), ms_per_step=f32[head])
        res["graph"] = {"train_samples_per_s": world * args.batch / (1e-3 * f32[head]), "ms_per_step": f32[head], "exchange": head}
        res["best_samples_per_s"] = world * args.batch / min(dt_eager, 1e-3 * timed[fastest])
        if emit is not None and rank == 0:
            emit(res)
        del graphs
        sweep = exchange_sweep(tr.reducer, dev, dist, iters=max(1, min(5, steps)))
        alone = sweep.get("%s_%s" % (tr.reducer.algo, tr.reducer.payload))
        res["exchange"].update(exchange_alone_ms=sweep, allreduce_ms=alone,
                               fastest=min((k for k, v in sweep.items() if isinstance(v, float)), key=sweep.get, default=None))
        if isinstance(alone, float) and alone > 0:
            res["exchange"].update(overlapped_fraction_eager=max(0.0, min(1.0, 1.0 - exposed / alone)),
                                   overlapped_fraction_graph=max(0.0, min(1.0, 1.0 - exposed_g / alone)))
    graphs = None
    del tr, model, batches
    torch.cuda.empty_cache()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return res if rank == 0 else None


def train_leg_subprocess(args, world=1, dist=None):
    """The training leg in its own process (its 200 M-parameter model, graphs and pools are gone when it returns; a failure
    in this secondary leg cannot take the headline line with it).  Default runtime settings: the captured step holds
    kernel nodes only (gridmm_amd/train_graph.py).  With several
    ranks EVERY rank starts its child (same RANK / LOCAL_RANK / WORLD_SIZE, the next master port: the children form their
    own process group); rank 0's child prints the JSON."""
    import subprocess
    env = dict(os.environ)
    if world > 1:
        # the children's own rendezvous: a port rank 0 PROBED free (not "parent port + 1", which may be taken), agreed on
        # through the parent's process group
        port = [_free_port() if int(os.environ.get("RANK", "0")) == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(port, src=0)
        else:
            port = [int(os.environ.get("MASTER_PORT", "29500")) + 1]
        env["MASTER_PORT"] = str(int(port[0]))
        for k in [k for k in env if k.startswith("TORCHELASTIC_")]:
            env.pop(k)        # (TORCHELASTIC_USE_AGENT_STORE would make the children look for the launcher's store on the new port)
    limit = float(os.environ.get("GRIDMM_BENCH_TRAIN_TIMEOUT", "420"))
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--train-leg-only", "--batch", str(args.batch)],
                           env=env, capture_output=True, text=True, timeout=limit)
    except subprocess.TimeoutExpired as e:
        # the child prints its step timings before the per-algorithm exchange sweep: keep what it got to
        part = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        lines = [l for l in part.strip().splitlines() if l.startswith("{")]
        if int(os.environ.get("RANK", "0")) != 0:
            return None
        if lines:
            return dict(json.loads(lines[-1]), note="the training leg was cut after %.0f s (exchange sweep unfinished)" % limit)
        return {"error": "the training leg timed out after %.0f s" % limit}
    if int(os.environ.get("RANK", "0")) != 0:
        return None
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:      # a secondary key: reported in the line, the headline measurement stands
        sys.stderr.write("bench.py: the training leg failed:\n" + r.stdout[-2000:] + r.stderr[-4000:] + "\n")
        return {"error": "the training leg failed (rc %d): %s" % (r.returncode, r.stderr.strip().splitlines()[-1][:300] if r.stderr.strip() else "")}
    return json.loads(lines[-1])


def hbm_stream_peak(dev, gib=4):
    """Measured HBM streaming-READ rate (GB/s): gridmm_hbm_read_probe over a window of `gib` GiB (16x the 256 MiB Infinity
    Cache, so that nothing of a pass survives to the next), best of 5 timed passes after a warm-up."""
    import ctypes
    from gridmm_amd import _lib
    lib = _lib.load()
    n = int(gib) << 30
    buf = torch.empty(n, dtype=torch.uint8, device=dev)
    buf.view(torch.int32).fill_(1)
    out = torch.zeros(2048, dtype=torch.float32, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: _lib.check(lib.gridmm_hbm_read_probe(ctypes.c_void_p(buf.data_ptr()), n, ctypes.c_void_p(out.data_ptr()), st),   # noqa: E731
                              "gridmm_hbm_read_probe")
    call()
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record()
        torch.cuda.synchronize()
        best = max(best, n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del buf
    torch.cuda.empty_cache()
    return best


def roofline_leg(step, args, geom, L=80):
    """Per-kernel HIP-event timing over a few instrumented steps (events on the launch stream)."""
    from gridmm_amd import ops
    n = max(3, min(args.steps, 10))
    ops.TIMER = ops.KernelTimer()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    summ = ops.TIMER.summary()
    relaunch = getattr(ops.TIMER, "last_aggregate", None)
    ops.TIMER = None
    agg_ms = None
    if relaunch is not None:
        # Device-side duration of one aggregation launch (main kernel + the merge of split cells): 10 launches captured
        # in a hipGraph, replayed, HIP events on the replay stream.  The per-launch events of the eager steps above also
        # time the host's launch overhead between the two kernels.
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            relaunch()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):      # (a process group's watchdog thread may be polling events)
            for _ in range(10):
                relaunch()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        agg_ms = e0.elapsed_time(e1) / 50
    kern = {k: {"calls_per_step": v["calls"] / n, "ms_per_step": v["ms"] / n, "avg_us": 1e3 * v["ms"] / v["calls"]}
            for k, v in summ.items()}
    B, N, D = args.batch, geom.pts_per_obs * args.mem_steps, geom.feat_dim
    out = {"kernels": kern}
    if "linear" in summ:
        tf = summ["linear"]["work"] / (summ["linear"]["ms"] * 1e-3) / 1e12
        out["linear"] = {"bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                         "frac": tf / MFMA_BF16_PEAK_TF, "traffic": None,
                         "note": "algorithmic 2MNK flops of all GEMM launches / their summed HIP-event time; the "
                                 "3-term bf16 split issues 3x these flops on the matrix pipe"}
    if "grid_aggregate" in summ:
        # algorithmic bytes per launch: slab + perm + text fragments (hi+lo) + cell vectors out
        byts = B * (N * D * 2 + N * 4 + 2 * L * D * 2 + 196 * D * 4 + 196)
        per_launch_ms = agg_ms if agg_ms is not None else summ["grid_aggregate"]["ms"] / summ["grid_aggregate"]["calls"]
        gbs = byts / (per_launch_ms * 1e-3) / 1e9
        out["grid_aggregate"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": gbs / HBM_PEAK_GBS, "traffic": None, "bytes_per_launch": byts,
                                 "us_per_launch": 1e3 * per_launch_ms,
                                 "timing": "HIP events around hipGraph replays of the launch (aggregation kernel + merge "
                                           "kernel)" if agg_ms is not None else "HIP events around the eager launch"}
        try:       # the measured streaming-read peak of THIS box next to the specification
            pk = hbm_stream_peak(torch.device("cuda", torch.cuda.current_device()))
            out["grid_aggregate"].update(peak_measured=pk, frac_of_measured=gbs / pk,
                                         peak_measured_how="gridmm_hbm_read_probe: one pass over a 4 GiB window (16x the "
                                                           "Infinity Cache), 16-byte loads, best of 5")
        except Exception as e:
            out["grid_aggregate"]["peak_measured_error"] = repr(e)[:200]
    # HBM bytes per launch from the committed rocprofv3 PMC passes of this same workload (tools/collect_traffic.sh:
    # FETCH_SIZE and WRITE_SIZE in separate runs, (2*FETCH + WRITE) * 1024 with the gfx950 read-side correction)
    import glob
    cands = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_hbm_traffic.json")))
    tpath = cands[-1] if cands else ""               # the newest round's PMC passes
    if os.path.exists(tpath):
        t = json.load(open(tpath))
        c = t.get("config", {})
        if (int(c.get("batch", -1)), c.get("shape"), int(c.get("mem_steps", -1))) == (args.batch, args.shape, args.mem_steps):
            for k in ("linear", "grid_aggregate"):
                if k in out and k in t["kernels"]:
                    out[k]["traffic"] = t["kernels"][k]["hbm_bytes_per_launch"]
                    out[k]["traffic_source"] = "profiles/" + os.path.basename(tpath)
    # the dominant kernel among those with a roofline entry (at the BASELINE batch the GEMMs are 75 % of the step; with a
    # few episodes per rank and ranks sharing a device -- the 8-rank dry run of the tests -- the attention launches can
    # out-time them in the eager events, and they have no entry of their own: that made `roofline` come out without `frac`
    # once in ~10 runs)
    dom = max((k for k in ("linear", "grid_aggregate") if k in out), key=lambda k: summ[k]["ms"])
    out["dominant"] = dom
    return out


def cpu_baseline(model, eps, batch, args, geom):
    """The oracle (NumPy + torch-CPU port of the reference algorithm, incl. its 196-cell loop) timed on this
    host on a bounded sample of the same workload."""
    from oracle import navcmt_oracle as O, gridmap_oracle as G
    og = G.BASELINE if args.shape == "baseline" else G.NATIVE
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    keys = ("txt_embeds", "txt_masks", "gmap_img_embeds", "gmap_step_ids", "gmap_pos_fts", "gmap_masks",
            "gmap_visited_masks", "vp_img_embeds", "vp_pos_fts", "vp_masks", "vp_nav_masks")

    def run(b0, b1):
        refs = []
        for b in range(b0, b1):
            mem = G.GridMemory(og)
            for o in eps[b]:
                r = mem.step(o["depth"], o["feats"], o["x"], o["y"], o["heading"])
            refs.append(r)
        cb = {k: batch[k][b0:b1].cpu() for k in keys}
        cb.update(gmap_vpids=batch["gmap_vpids"][b0:b1], vp_cand_vpids=batch["vp_cand_vpids"][b0:b1],
                  vp_obj_masks=None, gmap_pair_dists=None,
                  grid_fts=[torch.from_numpy(r[0]) for r in refs], grid_map=[torch.from_numpy(r[1]) for r in refs],
                  gridmap_pos_fts=torch.from_numpy(np.stack([r[2] for r in refs])))
        with torch.no_grad():
            O.forward_navigation(sd, cb)

    best = None
    ncpu = os.cpu_count() or 1
    for k in sorted(set([1, min(8, ncpu)])):
        torch.set_num_threads(k)
        t0 = time.perf_counter()
        run(0, 1)
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (k, dt)
    k = best[0]
    torch.set_num_threads(k)
    done, t0 = 0, time.perf_counter()
    chunk = 2
    while time.perf_counter() - t0 < args.cpu_seconds:      # ~10-30 s of CPU work; wraps around the episode batch
        b0 = done % (len(eps) - chunk + 1)
        run(b0, b0 + chunk)
        done += chunk
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "steps/s", "cores": k, "kind": "port",
            "sample": "%d episode-steps of the same workload (N=%d points x %d-D, L=80, full-size model), "
                      "oracle/navcmt_oracle.py + oracle/gridmap_oracle.py on %d torch thread(s), %.1f s"
                      % (done, geom.pts_per_obs * args.mem_steps, geom.feat_dim, k, dt)}


def torch_gpu_baseline(model, batch, mem, args):
    """The reference-style PyTorch path on this GPU: the op-for-op oracle (196-cell python loop and all) run with
    stock torch ops on cuda -- the '>= 5x' comparator of the north star.  Bounded: 1 warm-up + 2 timed calls."""
    from oracle import navcmt_oracle as O
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    fts, gmaps, pos = mem.as_reference_obs()
    b = dict(batch, grid_fts=fts, grid_map=gmaps, gridmap_pos_fts=pos, grid_memory=None)
    with torch.no_grad():
        O.forward_navigation(sd, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 2
        for _ in range(n):
            O.forward_navigation(sd, b)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    return {"value": args.batch / dt, "unit": "steps/s", "kind": "port-on-gpu",
            "sample": "forward('navigation') only (grid map prebuilt), B=%d, fp32 stock torch ops, %.3f s/call"
                      % (args.batch, dt)}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) started WITHOUT a launcher: re-exec under torch.distributed.run, one rank per
    device -- what the reference's own scripts do (map_nav_src/scripts/run_r2r.sh:65, pretrain_src/run_r2r.sh:6-8:
    torch.distributed.launch --nproc_per_node).  The parent only forwards the children's output (rank 0's JSON line)
    and exit code; a one-rank line for --gpus N cannot be printed."""
    import subprocess
    share = bool(os.environ.get("GRIDMM_BENCH_SHARE_GPU"))
    have = torch.cuda.device_count()
    if have < args.gpus and not share:
        raise SystemExit("bench.py: --gpus %d but this node exposes %d device(s)" % (args.gpus, have))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    share = bool(os.environ.get("GRIDMM_BENCH_SHARE_GPU"))   # test hook: N ranks on one GPU (gloo for the timing collectives)
    if share:
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.train_leg_only:
        res = train_leg(args, dev, emit=lambda r: print(json.dumps(r), flush=True))
        if res is not None:
            print(json.dumps(res), flush=True)
        return
    dist = None
    if world > 1 or _dist_forced():
        # (GRIDMM_DIST_FORCE=1 with one rank: the multi-rank code of THIS leg -- captures next to a live communicator, barriers
        # and the max-over-ranks reduction over RCCL -- on a one-GPU box; tests/test_hip_dist.py)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist  # RCCL ("nccl" backend on ROCm)
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, "
                         "or plainly as `python bench.py --gpus %d`, which starts the ranks itself)"
                         % (args.gpus, world, args.gpus, args.gpus))

    model, batch, mem, eps, step, eager_step, geom = build_workload(args, dev)
    dt = time_steps(step, args.steps, args.warmup, dist)
    n_gpus = world
    value = n_gpus * args.batch * args.steps / dt
    check = check_replay(step, eager_step)      # the timed (replayed) step must reproduce the eager launches

    out = {
        "metric": METRIC, "value": value, "unit": "steps/s", "n_gpus": n_gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "R2R fine-tune batch=%d/GPU, %d views x %d patches x %dD, 14x14 grid, memory depth "
                               "t=%d (N=%d points), L=80, G=20, V=37, full-size GlocalTextPathNavCMT (random init): "
                               "fill_gridmap + forward('navigation')"
                               % (args.batch, geom.n_views, geom.patches ** 2, geom.feat_dim, args.mem_steps,
                                  geom.pts_per_obs * args.mem_steps),
                   "global_batch": args.batch * n_gpus, "parallelism": "dp%d (episode sharding, no step-path collective)" % n_gpus,
                   "launch": "eager" if args.eager else "hipGraph replay; per step on the host: pose/heading floats and the fused-logit index maps (H2D into static buffers)",
                   "gemm": "MFMA bf16 16x16x32, 3-term split (hi*hi+lo*hi+hi*lo), fp32 accumulate",
                   "attention": "MFMA bf16 16x16x32, 3-term split, fp32 softmax", "slab": "fp16, relevance on MFMA f16 (text hi+lo)",
                   "grid_proj": "post-reduction shortcut (declared, SURVEY 8d): grid_proj runs on the 196 reduced cell vectors "
                                "(W sum_j a_j x_j + b, sum_j a_j = 1) instead of on all N points as vilmodel.py:799 writes it -- "
                                "-5.4 GFLOP per episode-step, difference < 1e-6; the roofline flop count (2MNK of the launched "
                                "GEMMs) is the post-reduction count",
                   "instruction_side": "recomputed every step like the reference (the cached form is the secondary key "
                                       "instruction_cache)",
                   "relevance": ("one pass over the slab (relevance + per-cell softmax sums in one kernel)"
                                 if geom.feat_dim != 768 else
                                 "D = 768: relevance pass + accumulation pass; the relevance of the points appended at earlier "
                                 "steps is %s" % ("recomputed every step like the reference (--no-relevance-cache)"
                                                  if args.no_relevance_cache else
                                                  "kept next to the device-resident slab (it depends on the slab row and the "
                                                  "instruction only; bit-identical), so a step computes the new observation's "
                                                  "values and reads the slab once -- --no-relevance-cache recomputes all"))},
    }
    out["replay_check"] = check
    if getattr(step, "graph", None) is not None and step.graph.n_nodes is not None:
        out["graph_nodes"] = step.graph.n_nodes       # nodes of one replayed step (hipGraphGetNodes)
    if not args.no_depth_legs and not args.eager and args.mem_steps == 1:
        # the memory deepens as an episode proceeds; the headline is t = 1, these are the same step at
        # t = 5 and t = 15 (re-binning and aggregation walk 5x / 15x the points)
        for t in (5, 15):
            sec = extra_depth_leg(args, dev, dist, t, max(10, args.steps))    # (10-step timings of these legs moved by 10 % between runs)
            out["t%d" % t] = {"value": n_gpus * args.batch / sec, "unit": "steps/s", "ms_per_step": 1e3 * sec,
                              "mem_steps": t, "points": geom.pts_per_obs * t}
    if not args.no_depth_legs and not args.eager and args.mem_steps == 1 and n_gpus == 1:
        out["sparse_map"] = sparse_map_leg(args, dev, dist, max(5, args.steps // 2))
    if rank == 0 and not args.no_depth_legs and not args.eager and n_gpus == 1:
        out.update(config_legs(args, dev, max(5, args.steps // 2)))
        out["instruction_cache"] = instruction_cache_leg(args, dev, max(5, args.steps // 2), out["ms_per_step"])
        if args.batch == 32 and args.mem_steps == 1:
            out["larger_batches"] = larger_batch_leg(args, dev, max(5, args.steps // 2))
    def _rollout():
        try:
            out["rollout"] = rollout_leg(args, dev)
        except Exception as e:      # a secondary key: reported in the line, the headline measurement stands
            out["rollout"] = {"error": repr(e)[:300]}
    if rank == 0 and not args.no_depth_legs and not args.eager and n_gpus == 1:
        _rollout()
    if rank == 0 and not args.no_depth_legs and not args.eager and n_gpus == 1 and not args.no_train_leg:
        try:
            out["finetune"] = finetune_leg(args, dev)
        except Exception as e:      # a secondary key: reported in the line, the headline measurement stands
            out["finetune"] = {"error": repr(e)[:300]}
    if not args.no_train_leg:
        # config 3's shape: the pre-training step on EVERY rank with the RCCL gradient exchange (whole-job samples/s)
        tl = train_leg_subprocess(args, n_gpus, dist)
        if rank == 0:
            out["train"] = tl
    if rank == 0 and not args.no_depth_legs and not args.eager and n_gpus > 1:
        # N > 1: rank 0's end-to-end rollout (one GPU's worth; the other ranks wait at the final barrier), after the legs
        # that need every rank
        _rollout()
        if isinstance(out.get("rollout"), dict):
            out["rollout"]["note"] = "rank 0 only (one GPU); the other ranks idle"
    if rank == 0 and n_gpus == 1 and not args.no_producer_leg:
        out["vlnce_with_producer"] = producer_leg(args, dev)
    if rank == 0 and not args.no_roofline:
        rl = roofline_leg(eager_step, args, geom)   # per-launch HIP events need eager launches
        dom = rl["dominant"]
        out["roofline"] = dict(rl.get(dom, {}), kernel=dom)
        for k in ("linear", "grid_aggregate"):
            if k != dom and k in rl:
                out["roofline_" + k] = rl[k]
        out["kernels"] = rl["kernels"]
    if rank == 0:
        if not args.no_torch_gpu_baseline and n_gpus == 1:
            out["torch_gpu_baseline"] = torch_gpu_baseline(model, batch, mem, args)
        if not args.no_cpu_baseline:
            # (also in the N > 1 line -- a SCALE line carries roofline AND cpu_baseline; timed on rank 0's host cores after
            # every timed leg, the other ranks wait at the barrier below)
            out["cpu_baseline"] = cpu_baseline(model, eps, batch, args, geom)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
