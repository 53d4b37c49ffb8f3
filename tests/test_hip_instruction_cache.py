"""GPU: the per-episode instruction cache (GlocalTextPathNavCMT.instruction_cache) and the two-buffer context of
gridmm_attention_rows_seg.  The reference recomputes text_proj, the instruction's K / V of the grid / text layer and the
instruction rows of the local encoder's K / V at every step (map_nav_src/models/vilmodel.py:793, 841-853) from a txt_embeds
that is constant over the episode: row-wise projections of constant rows, so the cached step must give the SAME BITS."""
import argparse
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
KEYS = ("global_logits", "local_logits", "grid_logits", "fused_logits")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda")


@pytest.mark.parametrize("Sq,S1,S2", [(57, 216, 80), (57, 100, 37), (216, 33, 64), (20, 7, 1)])
def test_attention_over_two_context_buffers_equals_the_concatenation(dev, Sq, S1, S2):
    from gridmm_amd import ops
    B, H = 3, 768
    g = torch.Generator().manual_seed(Sq + S1 + S2)
    q = ops.split_rows(torch.randn(B, Sq, H, generator=g).to(dev))
    kv1 = ops.split_rows(torch.randn(B, S1, 4 * H, generator=g).to(dev))      # K at column 2H, V at 3H (a shared K/V buffer)
    kv2 = ops.split_rows(torch.randn(B, S2, 2 * H, generator=g).to(dev))      # K at 0, V at H
    cat_hi = torch.cat([kv1.hi[..., 2 * H:], kv2.hi], 1).contiguous()
    cat_lo = torch.cat([kv1.lo[..., 2 * H:], kv2.lo], 1).contiguous()
    lens = torch.tensor([S1 + S2, max(1, S1 - 3), S1 + max(1, S2 // 2)])
    mask = (torch.arange(S1 + S2)[None] < lens[:, None]).to(dev)
    sl = lambda a, c0: (a.hi[..., c0:c0 + H], a.lo[..., c0:c0 + H])
    want = ops.attention_rows((q.hi, q.lo), (cat_hi[..., :H], cat_lo[..., :H]), (cat_hi[..., H:], cat_lo[..., H:]), mask,
                              want_f32=True)
    got = ops.attention_rows((q.hi, q.lo), sl(kv1, 2 * H), sl(kv1, 3 * H), mask, want_f32=True, k2=sl(kv2, 0), v2=sl(kv2, H))
    torch.cuda.synchronize()
    assert torch.equal(got.f32, want.f32) and torch.equal(got.hi, want.hi) and torch.equal(got.lo, want.lo)
    assert float(want.f32.abs().max()) > 0.01


def _args(**kw):
    d = dict(batch=32, shape="baseline", mem_steps=1, eager=False)
    d.update(kw)
    return argparse.Namespace(**d)


@pytest.mark.parametrize("B", [32, 3])
def test_cached_step_gives_the_same_bits_as_the_recomputing_step(dev, B):
    """Full-size model at the bench configuration: eager step, eager step with the cache, captured step with the cache."""
    import bench
    model, batch, mem, eps, step, eager_step, geom = bench.build_workload(_args(batch=B), dev, instruction_cache=True)
    want = {k: v.clone() for k, v in eager_step().items() if k in KEYS}
    ic = model.instruction_cache(batch["txt_embeds"], batch["txt_masks"])
    # (eager_step restores the memory, appends the observation and calls the model; the cached variant of the same call:)
    from gridmm_amd import ops
    with ops_timer_off():
        got_graph = {k: v.clone() for k, v in step().items() if k in KEYS}
    mem_batch = dict(batch, instruction_cache=ic, fusion_maps=step.graph.batch["fusion_maps"])
    got_eager = {k: v.clone() for k, v in model("navigation", mem_batch).items() if k in KEYS}
    torch.cuda.synchronize()
    for k in KEYS:
        assert torch.equal(got_graph[k], want[k]), k
        assert torch.equal(got_eager[k], want[k]), k
    assert step.graph.n_nodes is None or step.graph.n_nodes < 84      # five launches fewer than the recomputing step


class ops_timer_off:
    def __enter__(self):
        from gridmm_amd import ops
        self.keep, ops.TIMER = ops.TIMER, None
    def __exit__(self, *a):
        from gridmm_amd import ops
        ops.TIMER = self.keep


def test_navigation_graphs_fill_the_cache_once_per_instruction_tensor(dev):
    """graph.NavigationGraphs (the rollout's path): varlen map sequences, several shape keys; one fill per txt_embeds tensor,
    logits equal to the uncached NavigationGraphs bit for bit."""
    import bench
    from gridmm_amd.graph import NavigationGraphs
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT
    model, batch, mem, eps, step, eager_step, geom = bench.build_workload(_args(batch=8, eager=True), dev)
    eager_step()
    model.varlen_buckets = GlocalTextPathNavCMT.DEFAULT_BUCKETS
    try:
        a, b = NavigationGraphs(model), NavigationGraphs(model)
        b.use_instruction_cache = False
        outs = []
        for ng in (a, b):
            res = []
            for _ in range(3):
                res.append({k: v.clone() for k, v in ng(dict(batch)).items() if k in KEYS})
            outs.append(res)
        torch.cuda.synchronize()
        for ra, rb in zip(*outs):
            for k in KEYS:
                assert torch.equal(ra[k], rb[k]), k
        assert a.icache_fills == 1 and b.icache_fills == 0
        batch2 = dict(batch, txt_embeds=batch["txt_embeds"] * 1.5)            # a new episode batch: new tensor, refill
        ra = {k: v.clone() for k, v in a(batch2).items() if k in KEYS}
        rb = {k: v.clone() for k, v in b(batch2).items() if k in KEYS}
        torch.cuda.synchronize()
        assert a.icache_fills == 2 and all(torch.equal(ra[k], rb[k]) for k in KEYS)
        assert not torch.equal(ra["fused_logits"], outs[0][0]["fused_logits"])
    finally:
        model.varlen_buckets = None


def test_language_graphs_equal_the_eager_encoder_and_return_fresh_tensors(dev):
    """graph.LanguageGraphs (forward('language') of a rollout from a hipGraph per (B, L)): the same bits as the eager call for
    every shape, a FRESH tensor per call (the per-episode caches downstream recognise a new instruction by the tensor's
    identity), least-recently-used eviction beyond `max_graphs`, re-capture after an in-place weight update."""
    from gridmm_amd.graph import LanguageGraphs
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    torch.manual_seed(0)
    model = GlocalTextPathNavCMT(default_config(num_l_layers=2, num_pano_layers=1, num_x_layers=1, intermediate_size=256,
                                                vocab_size=500)).eval().to(dev)
    lg = LanguageGraphs(model, max_graphs=2)
    g = torch.Generator().manual_seed(1)
    seen = []
    for B, L in ((3, 17), (3, 40), (3, 17), (5, 17), (3, 40), (3, 17)):
        ids = torch.randint(1, 500, (B, L), generator=g).to(dev)
        lens = torch.randint(1, L + 1, (B,), generator=g)
        lens[0] = L
        masks = (torch.arange(L)[None] < lens[:, None]).to(dev)
        lg.validate()
        got = lg({"txt_ids": ids, "txt_masks": masks})
        with torch.no_grad():
            want = model("language", {"txt_ids": ids, "txt_masks": masks})
        assert torch.equal(got, want), (B, L)
        assert all(got.data_ptr() != t.data_ptr() for t in seen)
        seen.append(got)
    assert len(lg.graphs) == 2 and lg.replays == 6 and lg.captures == 5      # (3,17) hit once; evicted and re-captured later
    with torch.no_grad():
        model.embeddings.word_embeddings.weight.mul_(1.5)
    lg.validate()
    assert len(lg.graphs) == 0
    got = lg({"txt_ids": ids, "txt_masks": masks})
    with torch.no_grad():
        assert torch.equal(got, model("language", {"txt_ids": ids, "txt_masks": masks}))
