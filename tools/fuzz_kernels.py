"""Randomised shape sweep of the hot-path kernels on the GPU: every case is checked against an fp64 / oracle reference AND
run twice for bit-equality (a race shows as run-to-run variation long before it shows as a wrong value).
usage: python tools/fuzz_kernels.py [--cases 40] [--seed 0] [--only gemm,attention,layernorm,aggregate,nav,linear_bwd,attn_bwd,agg_bwd]
Prints one line per failing case and a summary; exit code 1 on any failure."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from gridmm_amd import ops  # noqa: E402

DEV = torch.device("cuda")
FAILS = []


def fail(kind, case, msg):
    FAILS.append((kind, case, msg))
    print("FAIL %-10s %s : %s" % (kind, case, msg), flush=True)


def fuzz_gemm(rs, n):
    for _ in range(n):
        M = int(rs.choice([1, 7, 57, 64, 129, 500, 1824, 2560, 4097, 6912]))
        K = int(rs.choice([32, 64, 96, 512, 768, 1536, 3072]))
        N = int(rs.choice([4, 12, 64, 196, 512, 768, 1000, 2304, 3072]))
        act = int(rs.choice([ops.ACT_NONE, ops.ACT_GELU, ops.ACT_RELU]))
        res = bool(rs.rand() < 0.4)
        g = torch.Generator().manual_seed(int(rs.randint(1 << 30)))
        x = torch.randn(M, K, generator=g)
        w, b = torch.randn(N, K, generator=g) / np.sqrt(K), torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g) if res else None
        pw = ops.PackedLinear(w.to(DEV), b.to(DEV))
        xa = ops.split_rows(x.to(DEV))
        outs = []
        for _rep in range(2):
            y = ops.linear(xa, pw, act=act, residual=None if r is None else r.to(DEV), want_f32=True, want_planes=True)
            torch.cuda.synchronize()
            outs.append((y.f32.clone(), y.hi.clone(), y.lo.clone()))
        ref = x.double() @ w.double().t() + b.double()
        if act == ops.ACT_GELU:
            ref = torch.nn.functional.gelu(ref)
        elif act == ops.ACT_RELU:
            ref = torch.relu(ref)
        if r is not None:
            ref = ref + r.double()
        err = float((outs[0][0].double().cpu() - ref).abs().max())
        case = (M, N, K, act, res)
        if err > 3e-4 * max(1.0, float(ref.abs().max())):
            fail("gemm", case, "max abs err %.3e" % err)
        if not all(torch.equal(a, b2) for a, b2 in zip(outs[0], outs[1])):
            fail("gemm", case, "run-to-run variation")
        rec = outs[0][1].float() + outs[0][2].float()
        if float((rec - outs[0][0]).abs().max()) > 2.0 ** -15 * max(1e-6, float(outs[0][0].abs().max())):
            fail("gemm", case, "planes do not reconstruct the fp32 output")


def fuzz_attention(rs, n):
    for _ in range(n):
        B = int(rs.choice([1, 2, 5]))
        Sq = int(rs.choice([1, 5, 16, 57, 90, 216, 300]))
        Sk = int(rs.choice([1, 31, 32, 57, 80, 216, 296, 513, 700, 1100]))
        g = torch.Generator().manual_seed(int(rs.randint(1 << 30)))
        q, kv = torch.randn(B, Sq, 768, generator=g), torch.randn(B, Sk, 1536, generator=g)
        mask = torch.rand(B, Sk, generator=g) < 0.8
        mask[:, 0] = True
        if rs.rand() < 0.3 and Sk >= 2:
            mask[0, Sk // 2:] = False
        qa, ka = ops.split_rows(q.to(DEV)), ops.split_rows(kv.to(DEV))
        ksl, vsl = (ka.hi[..., :768], ka.lo[..., :768]), (ka.hi[..., 768:], ka.lo[..., 768:])
        outs = []
        for _rep in range(2):
            o = ops.attention_rows((qa.hi, qa.lo), ksl, vsl, mask.to(DEV), want_f32=True, want_planes=True)
            torch.cuda.synchronize()
            outs.append((o.f32.clone(), o.hi.clone(), o.lo.clone()))

        def heads(t):
            return t.reshape(B, -1, 12, 64).permute(0, 2, 1, 3).double()
        s = heads(q) @ heads(kv[..., :768]).transpose(-1, -2) / 8.0
        s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
        ref = (torch.softmax(s, -1) @ heads(kv[..., 768:])).permute(0, 2, 1, 3).reshape(B, Sq, 768)
        err = float((outs[0][0].double().cpu() - ref).abs().max())
        case = (B, Sq, Sk)
        if not (err < 3e-4):
            fail("attention", case, "max abs err %.3e" % err)
        if not all(torch.equal(a, b2) for a, b2 in zip(outs[0], outs[1])):
            fail("attention", case, "run-to-run variation")


def fuzz_layernorm(rs, n):
    for _ in range(n):
        M = int(rs.choice([1, 3, 57, 1824, 2049, 6912]))
        H = 768
        g = torch.Generator().manual_seed(int(rs.randint(1 << 30)))
        x, r = torch.randn(M, H, generator=g) * 3, torch.randn(M, H, generator=g)
        gm, bt = torch.randn(H, generator=g), torch.randn(H, generator=g)
        use_r = bool(rs.rand() < 0.5)
        outs = []
        for _rep in range(2):
            y = ops.layernorm(x.to(DEV), gm.to(DEV), bt.to(DEV), 1e-12, residual=r.to(DEV) if use_r else None, want_planes=True)
            torch.cuda.synchronize()
            outs.append((y.f32.clone(), y.hi.clone(), y.lo.clone()))
        ref = torch.nn.functional.layer_norm((x + r if use_r else x).double(), (H,), gm.double(), bt.double(), 1e-12)
        err = float((outs[0][0].double().cpu() - ref).abs().max())
        if err > 2e-5 * max(1.0, float(ref.abs().max())):
            fail("layernorm", (M, use_r), "max abs err %.3e" % err)
        if not all(torch.equal(a, b2) for a, b2 in zip(outs[0], outs[1])):
            fail("layernorm", (M, use_r), "run-to-run variation")


def fuzz_aggregate(rs, n):
    import test_hip_aggregate as T
    from gridmm_amd.grid_memory import pack_reference_lists
    for _ in range(n):
        D = int(rs.choice([256, 512, 768]))
        L = int(rs.choice([5, 16, 33, 48, 80, 96, 97, 120, 200, 256, 257, 300, 512]))
        kind = str(rs.choice(["sparse", "crowded", "blocks"]))
        B = int(rs.choice([1, 2, 4]))
        npts = [int(rs.choice([0, 1, 31, 32, 33, 200, 588, 1500, 4000, 9000])) for _ in range(B)]
        n_chunks = rs.choice([None, 1, 2, 4, 8, 24, 64])
        n_chunks = None if n_chunks is None else int(n_chunks)
        seed = int(rs.randint(1 << 30))
        rng, g = np.random.default_rng(seed), torch.Generator().manual_seed(seed)
        fts, maps = [], []
        for npt in npts:
            fts.append((torch.randn(npt, D, generator=g) * 0.5).half())
            maps.append(torch.from_numpy(T._episode_ids(kind, npt, rng).astype(np.int64)) if npt else torch.zeros(0, dtype=torch.int64))
        text = torch.randn(B, L, D, generator=g) * 0.3
        case = (D, L, kind, npts, n_chunks)
        if max(npts) == 0:
            continue
        slab, perm, cs = pack_reference_lists([f.to(DEV) for f in fts], [m.to(DEV).double() for m in maps])
        outs = []
        for _rep in range(2):
            cells, occ, rel, amax = ops.grid_aggregate(slab, perm, cs, ops.text_fragments(text.to(DEV)), L, n_chunks=n_chunks,
                                                       want_relevance=True, want_amax=True)
            torch.cuda.synchronize()
            outs.append((cells.clone(), occ.clone(), rel.clone()))
        if not all(torch.equal(a, b2) for a, b2 in zip(outs[0], outs[1])):
            fail("aggregate", case, "run-to-run variation (rc %d)" % ops.LAST_AGGREGATE_RC)
        for b in range(B):
            ref_cells, ref_occ, w = T._ref(fts[b], maps[b], text[b], L)
            if not torch.equal(outs[0][1][b].cpu(), ref_occ):
                fail("aggregate", case, "occupancy differs (episode %d)" % b)
            err = float((outs[0][0][b].double().cpu() - ref_cells).abs().max())
            if not (err < 3e-5):
                fail("aggregate", case, "episode %d max abs err %.3e (rc %d)" % (b, err, ops.LAST_AGGREGATE_RC))


def fuzz_aggregate_incremental(rs, n):
    """gridmm_grid_aggregate_incremental over random episodes: every step appends an observation to a random subset of the
    episodes, re-draws every cell, and must equal gridmm_grid_aggregate on the same state bit for bit (cells, occupancy,
    relevance by sorted position); sometimes the instruction changes (the caller clears `valid`) or a step takes the full form."""
    import test_hip_aggregate_incremental as T
    for _ in range(n):
        D = int(rs.choice([256, 512, 768]))
        L = int(rs.choice([5, 16, 20, 32, 80, 97, 120, 200, 300])) if D != 768 else int(rs.choice([5, 16, 40, 80, 96, 120, 200]))
        if not ops.two_pass_aggregation(D, L):
            continue
        B = int(rs.choice([1, 2, 4]))
        n_new = int(rs.choice([31, 64, 200, 588, 1000, 2100]))
        steps = int(rs.choice([2, 3, 5]))
        n_chunks = rs.choice([None, 1, 4, 8, 24])
        n_chunks = None if n_chunks is None else int(n_chunks)
        seed = int(rs.randint(1 << 30))
        rng, g = np.random.default_rng(seed), torch.Generator().manual_seed(seed)
        case = (D, L, B, n_new, steps, n_chunks, seed)
        cap = n_new * steps
        slab = torch.zeros(B, cap, D, dtype=torch.float16, device=DEV)
        frag = ops.text_fragments((torch.randn(B, L, D, generator=g) * 0.3).to(DEV))
        st = T._state(B, cap)
        n_host = np.zeros(B, np.int64)
        n_pts = torch.zeros(B, dtype=torch.int32, device=DEV)
        invalid = rng.random((B, cap)) < 0.1
        try:
            for t in range(steps):
                act = rng.random(B) < 0.8 if t else np.ones(B, bool)
                for b in np.nonzero(act)[0]:
                    slab[b, n_host[b]:n_host[b] + n_new] = (torch.randn(n_new, D, generator=g) * 0.5).half().to(DEV)
                    n_host[b] += n_new
                n_pts.copy_(torch.from_numpy(n_host.astype(np.int32)))
                ids = rng.integers(0, 196, size=(B, cap))
                ids[invalid] = -1
                if rng.random() < 0.2:
                    frag = ops.text_fragments((torch.randn(B, L, D, generator=g) * 0.3).to(DEV))
                    st["valid"].zero_()
                T._check(slab, torch.from_numpy(ids.astype(np.int16)).to(DEV), n_pts, frag, L, st,
                         torch.from_numpy(act.astype(np.uint8)).to(DEV), n_new, n_chunks, full=bool(rng.random() < 0.15))
        except AssertionError as e:
            fail("agg_inc", case, "differs from gridmm_grid_aggregate at step %d: %r" % (t, e))


def fuzz_nav(rs, n):
    """Whole forward('navigation') on a reduced model against the CPU oracle, random batch / graph / view / instruction
    sizes, list-form grid inputs, with and without varlen buckets."""
    from gridmm_amd import synthetic as S
    from gridmm_amd.vilmodel import GlocalTextPathNavCMT, default_config
    from oracle import gridmap_oracle as G, navcmt_oracle as O
    from oracle.ref_harness import det_tensor
    cfg = default_config(num_l_layers=1, num_pano_layers=1, num_x_layers=2, intermediate_size=256, vocab_size=1000)
    model = GlocalTextPathNavCMT(cfg).eval()
    sd = {k: (det_tensor(k, v.shape, 3) if v.dtype.is_floating_point else v) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    model.to(DEV)
    for _ in range(n):
        B = int(rs.choice([1, 2, 3, 5]))
        L = int(rs.choice([6, 20, 40, 80, 130]))
        Gn = int(rs.choice([8, 10, 12, 30, 60]))
        V1 = int(rs.choice([4, 10, 37, 58]))
        T = int(rs.choice([1, 2, 4]))
        seed = int(rs.randint(1 << 30))
        r2 = np.random.RandomState(seed)
        mems = [G.GridMemory(G.NATIVE) for _ in range(B)]
        ref = None
        for t in range(T):
            eps = [S.make_observations(r2, S.NATIVE, 1, feat_scale=0.35)[0] for _ in range(B)]
            ref = [mems[b].step(eps[b]["depth"], eps[b]["feats"], eps[b]["x"], eps[b]["y"], eps[b]["heading"]) for b in range(B)]
        batch = S.make_nav_batch(r2, B, L=L, G=Gn, n_visited=max(2, (Gn - 4) // 3), V1=V1, n_cand=min(3, V1 - 1), min_len=min(5, L))
        cpu = dict(batch, grid_fts=[torch.from_numpy(r[0]) for r in ref], grid_map=[torch.from_numpy(r[1]) for r in ref],
                   gridmap_pos_fts=torch.from_numpy(np.stack([r[2] for r in ref])))
        case = (B, L, Gn, V1, T)
        with torch.no_grad():
            want = O.forward_navigation(sd, cpu)
            gpu = dict(S.batch_to(batch, DEV), grid_fts=[x.to(DEV) for x in cpu["grid_fts"]],
                       grid_map=[x.to(DEV) for x in cpu["grid_map"]], gridmap_pos_fts=cpu["gridmap_pos_fts"].to(DEV))
            for buckets in (None, GlocalTextPathNavCMT.DEFAULT_BUCKETS):
                model.varlen_buckets = buckets
                a = model("navigation", gpu)
                b2 = model("navigation", gpu)
                torch.cuda.synchronize()
                for k in ("global_logits", "local_logits", "fused_logits", "grid_logits"):
                    w, x, y = want[k], a[k].cpu(), b2[k].cpu()
                    f = torch.isfinite(w)
                    if not torch.equal(f, torch.isfinite(x)):
                        fail("nav", case, "%s: -inf placement differs (buckets %s)" % (k, buckets is not None))
                        continue
                    err = float((x[f] - w[f]).abs().max()) if f.any() else 0.0
                    if not (err < 3e-4):
                        fail("nav", case, "%s max abs err %.3e (buckets %s)" % (k, err, buckets is not None))
                    if not torch.equal(x, y):
                        fail("nav", case, "%s run-to-run variation (buckets %s)" % (k, buckets is not None))
        model.varlen_buckets = None


def _rel(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max() / (b.detach().double().abs().max() + 1e-30))


def fuzz_linear_bwd(rs, n):
    from gridmm_amd import autograd as ag
    for _ in range(n):
        lead = int(rs.choice([1, 2, 3]))
        M = int(rs.choice([1, 5, 33, 57, 130, 300, 700, 2304]))
        K = int(rs.choice([5, 7, 14, 32, 64, 768, 3072]))
        N = int(rs.choice([1, 32, 64, 96, 768, 1000, 3072]))
        bias, res = bool(rs.rand() < 0.7), bool(rs.rand() < 0.3)
        g = torch.Generator().manual_seed(int(rs.randint(1 << 30)))
        x0 = torch.randn(lead, M, K, generator=g)
        w0, b0 = torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g) * 0.1
        r0, dy = torch.randn(lead, M, N, generator=g), torch.randn(lead, M, N, generator=g).to(DEV)
        runs = []
        for _rep in range(2):
            x, w = x0.to(DEV).requires_grad_(), w0.to(DEV).requires_grad_()
            b = b0.to(DEV).requires_grad_() if bias else None
            r = r0.to(DEV).requires_grad_() if res else None
            y = ag.linear(x, w, b, r)
            y.backward(dy)
            torch.cuda.synchronize()
            runs.append([y.detach().clone()] + [t.grad.clone() for t in (x, w, b, r) if t is not None])
        xd, wd = x0.double().requires_grad_(), w0.double().requires_grad_()
        bd = b0.double().requires_grad_() if bias else None
        rd = r0.double().requires_grad_() if res else None
        yd = torch.nn.functional.linear(xd, wd, bd)
        if res:
            yd = yd + rd
        yd.backward(dy.double().cpu())
        want = [yd] + [t.grad for t in (xd, wd, bd, rd) if t is not None]
        case = (lead, M, K, N, bias, res)
        for i, (a, e) in enumerate(zip(runs[0], want)):
            if _rel(a.cpu(), e) > 5e-5:
                fail("linear_bwd", case, "tensor %d rel err %.2e" % (i, _rel(a.cpu(), e)))
        if not all(torch.equal(a, b2) for a, b2 in zip(runs[0], runs[1])):
            fail("linear_bwd", case, "run-to-run variation")


def fuzz_attention_bwd(rs, n):
    from gridmm_amd import autograd as ag

    def ref_att(q, k, v, kmask, heads):
        B, Sq, H = q.shape
        qh, kh, vh = (t.reshape(B, -1, heads, 64).transpose(1, 2) for t in (q, k, v))
        s_ = qh @ kh.transpose(-1, -2) / 8.0
        s_ = s_.masked_fill(~kmask[:, None, None, :], -float("inf"))
        return (torch.softmax(s_, -1) @ vh).transpose(1, 2).reshape(B, Sq, H)
    for _ in range(n):
        B, heads = int(rs.choice([1, 2, 3])), int(rs.choice([2, 12]))
        Sq = int(rs.choice([1, 17, 57, 80, 216, 300]))
        Sk = int(rs.choice([3, 45, 80, 216, 296, 350]))
        H = heads * 64
        g = torch.Generator().manual_seed(int(rs.randint(1 << 30)))
        lens = torch.randint(max(1, Sk // 3), Sk + 1, (B,), generator=g)
        lens[0] = Sk
        kmask = torch.arange(Sk)[None] < lens[:, None]
        q0, kv0 = torch.randn(B, Sq, H, generator=g), torch.randn(B, Sk, 2 * H, generator=g)
        dy = torch.randn(B, Sq, H, generator=g)
        runs = []
        for _rep in range(2):
            q, kv = q0.to(DEV).requires_grad_(), kv0.to(DEV).requires_grad_()
            y = ag.cross_attention(q, kv, kmask.to(DEV), heads, kv_col=0)
            y.backward(dy.to(DEV))
            torch.cuda.synchronize()
            runs.append([y.detach().clone(), q.grad.clone(), kv.grad.clone()])
        qd, kvd = q0.double().requires_grad_(), kv0.double().requires_grad_()
        yd = ref_att(qd, kvd[..., :H], kvd[..., H:], kmask, heads)
        yd.backward(dy.double())
        case = (B, heads, Sq, Sk)
        for i, (a, e) in enumerate(zip(runs[0], (yd, qd.grad, kvd.grad))):
            if _rel(a.cpu(), e) > 5e-5:
                fail("attn_bwd", case, "tensor %d rel err %.2e" % (i, _rel(a.cpu(), e)))
        if not all(torch.equal(a, b2) for a, b2 in zip(runs[0], runs[1])):
            fail("attn_bwd", case, "run-to-run variation")


def fuzz_attention_train(rs, n):
    """The attention of the differentiable path on the bf16 matrix pipe (round 5): projections that carry planes (q plain, k | v
    shifted by row 0 of the episode) -> gridmm_attention_rows_train / _bwd, against fp64 autograd of the same expression, on
    inputs with a LARGE common component in the context rows (what the shift exists for); every case twice (bit equality)."""
    from gridmm_amd import autograd as ag

    def ref_att(q, k, v, kmask, heads):
        B, Sq, H = q.shape
        qh, kh, vh = (t.reshape(B, -1, heads, 64).transpose(1, 2) for t in (q, k, v))
        s_ = qh @ kh.transpose(-1, -2) / 8.0
        s_ = s_.masked_fill(~kmask[:, None, None, :], -float("inf"))
        return (torch.softmax(s_, -1) @ vh).transpose(1, 2).reshape(B, Sq, H)
    for _ in range(n):
        B, heads = int(rs.choice([1, 2, 3])), int(rs.choice([2, 12]))
        Sq = int(rs.choice([1, 17, 57, 80, 216, 300]))
        Sk = int(rs.choice([3, 45, 80, 216, 296, 350]))
        same = bool(rs.randint(2))
        if same:
            Sk = Sq
        H = heads * 64
        g = torch.Generator().manual_seed(int(rs.randint(1 << 30)))
        lens = torch.randint(max(1, Sk // 3), Sk + 1, (B,), generator=g)
        lens[0] = Sk
        kmask = torch.arange(Sk)[None] < lens[:, None]
        common = float(rs.choice([0.0, 3.0, 10.0])) * torch.randn(1, 1, H, generator=g)
        x0 = torch.randn(B, Sq, H, generator=g) + (common if same else 0.0)
        c0 = x0 if same else torch.randn(B, Sk, H, generator=g) + common
        wq0 = torch.randn(3 * H if same else H, H, generator=g) * 0.03
        wkv0 = torch.randn(2 * H, H, generator=g) * 0.03
        bkv0 = torch.randn(2 * H, generator=g) * 0.1
        dy = torch.randn(B, Sq, H, generator=g)
        runs = []
        for _rep in range(2):
            x = x0.to(DEV).requires_grad_()
            wq, wkv, bkv = (torch.nn.Parameter(t.to(DEV)) for t in (wq0, wkv0, bkv0))
            if same:
                qkv = ag.linear(x, wq, None, out_planes=H)
                y = ag.self_attention(qkv, kmask.to(DEV), heads)
                outs = [x, wq]
            else:
                c = c0.to(DEV).requires_grad_()
                q = ag.linear(x, wq, None, out_planes=True)
                kv = ag.linear(c, wkv, bkv, out_planes=0)
                y = ag.cross_attention(q, kv, kmask.to(DEV), heads, kv_col=0)
                outs = [x, c, wq, wkv, bkv]
            y.backward(dy.to(DEV))
            torch.cuda.synchronize()
            runs.append([y.detach().clone()] + [t.grad.clone() for t in outs])
        xd = x0.double().requires_grad_()
        wqd, wkvd, bkvd = (t.double().requires_grad_() for t in (wq0, wkv0, bkv0))
        if same:
            qkvd = xd @ wqd.t()
            yd = ref_att(qkvd[..., :H], qkvd[..., H:2 * H], qkvd[..., 2 * H:], kmask, heads)
            exp_in = [xd, wqd]
        else:
            cd = c0.double().requires_grad_()
            kvd = cd @ wkvd.t() + bkvd
            yd = ref_att(xd @ wqd.t(), kvd[..., :H], kvd[..., H:], kmask, heads)
            exp_in = [xd, cd, wqd, wkvd, bkvd]
        yd.backward(dy.double())
        case = (B, heads, Sq, Sk, same)
        for i, (a, e) in enumerate(zip(runs[0], [yd] + [t.grad for t in exp_in])):
            if _rel(a.cpu(), e) > 5e-4:        # (inputs up to |x| ~ 10 through bf16x3 projections AND attention)
                fail("attn_train", case, "tensor %d rel err %.2e" % (i, _rel(a.cpu(), e)))
        if not all(torch.equal(a, b2) for a, b2 in zip(runs[0], runs[1])):
            fail("attn_train", case, "run-to-run variation")


def fuzz_aggregate_bwd(rs, n):
    from gridmm_amd import autograd as ag
    for _ in range(n):
        B, D = int(rs.choice([1, 2, 3])), int(rs.choice([256, 512, 768]))
        L = int(rs.choice([8, 20, 40, 80, 96, 112, 144, 200, 270, 300]))
        n_obs = int(rs.choice([1, 2, 4]))
        cap = 588 * n_obs
        r2 = np.random.RandomState(int(rs.randint(1 << 30)))
        n_pts = r2.randint(cap // 2, cap + 1, size=B)
        slab = torch.from_numpy((r2.standard_normal((B, cap, D)) * 0.35).astype(np.float16)).to(DEV)
        ids = r2.randint(-1, 196, size=(B, cap)).astype(np.int16)
        ids[:, :50] = r2.randint(0, 4, size=(B, 50))
        for b in range(B):
            ids[b, n_pts[b]:] = -1
        ids_t = torch.from_numpy(ids).to(DEV)
        perm = torch.empty(B, cap, dtype=torch.int32, device=DEV)
        cell_start = torch.empty(B, 198, dtype=torch.int32, device=DEV)
        ops.grid_sort_ids(ids_t, torch.from_numpy(n_pts.astype(np.int32)).to(DEV), perm, cell_start)
        text0 = torch.from_numpy(r2.standard_normal((B, L, D)).astype(np.float32)) * 0.3
        dcells = torch.from_numpy(r2.standard_normal((B, 196, D)).astype(np.float32)).to(DEV)
        runs = []
        for _rep in range(2):
            text = text0.to(DEV).requires_grad_()
            cells, occ = ag.grid_aggregate(text, slab, perm, cell_start)
            cells.backward(dcells)
            torch.cuda.synchronize()
            runs.append([cells.detach().clone(), text.grad.clone()])
        td = text0.to(DEV).double().requires_grad_()
        ref = torch.zeros(B, 196, D, dtype=torch.float64, device=DEV)
        for b in range(B):
            x = slab[b].double()
            w = (x @ td[b].t()).max(-1)[0]
            for c in range(196):
                sel = ids_t[b] == c
                if sel.any():
                    ref[b, c] = (torch.softmax(w[sel], 0)[:, None] * x[sel]).sum(0)
        ref.backward(dcells.double())
        case = (B, D, L, n_obs)
        if _rel(runs[0][0], ref) > 1e-4:
            fail("agg_bwd", case, "cells rel err %.2e" % _rel(runs[0][0], ref))
        if _rel(runs[0][1], td.grad) > 2e-3:
            fail("agg_bwd", case, "dtext rel err %.2e" % _rel(runs[0][1], td.grad))
        if not all(torch.equal(a, b2) for a, b2 in zip(runs[0], runs[1])):
            fail("agg_bwd", case, "run-to-run variation")


def fuzz_gridmap(rs, n):
    """Re-binning + stable sort of ragged histories (episodes skipped by `active` masks) for every slice count, against
    numpy's stable argsort of the cell ids the kernel itself wrote (the ids are pinned bit-exact on the oracle elsewhere)."""
    from gridmm_amd import synthetic
    from gridmm_amd.grid_memory import GridMemoryBatch
    for _ in range(n):
        geom = synthetic.BASELINE if rs.rand() < 0.5 else synthetic.NATIVE
        B, T = int(rs.choice([1, 2, 3, 5])), int(rs.choice([1, 2, 3, 6, 10]))
        mem = GridMemoryBatch(B, geom, max_steps=T, device=DEV)
        for k in range(T):
            depth = np.stack([rs.randint(0, 20000, size=geom.n_views * geom.patches ** 2).astype(np.uint16) for _ in range(B)])
            depth[rs.rand(*depth.shape) < 0.15] = 0
            active = rs.rand(B) < 0.75
            active[rs.randint(B)] = True
            feats = torch.zeros(B, geom.pts_per_obs, geom.feat_dim, dtype=torch.float16, device=DEV)
            mem.step(depth, feats, [(float(rs.uniform(-6, 6)), float(rs.uniform(-6, 6))) for _ in range(B)],
                     [float(rs.uniform(-7, 7)) for _ in range(B)], active=None if active.all() else active)
        n_pts = mem.n_pts.cpu().numpy()
        case = (geom.pts_per_obs, B, T, n_pts.tolist())
        ref = None
        for S_ in (1, 2, 4, 8, 16):
            ops.grid_bin(mem.hist_x, mem.hist_y, mem.hist_valid, mem.n_pts, mem.pose_d, mem.head_d, mem.half_len, mem.cell_id,
                         mem.perm, mem.cell_start, mem.flags, workspace=mem._bin_ws, slices=S_)
            torch.cuda.synchronize()
            ids, perm, cs = mem.cell_id.cpu().numpy(), mem.perm.cpu().numpy(), mem.cell_start.cpu().numpy()
            for b in range(B):
                nb = int(n_pts[b])
                key = np.where(ids[b, :nb] < 0, 196, ids[b, :nb]).astype(np.int64)
                want = np.argsort(key, kind="stable")
                counts = np.bincount(key, minlength=197)
                starts = np.concatenate([[0], np.cumsum(counts)])
                if not np.array_equal(perm[b, :nb], want):
                    fail("gridmap", case, "perm differs (S=%d, episode %d)" % (S_, b))
                if not np.array_equal(cs[b, :198], starts[:198]):
                    fail("gridmap", case, "cell_start differs (S=%d, episode %d)" % (S_, b))
            if ref is None:
                ref = ids.copy()
            elif not np.array_equal(ref, ids):
                fail("gridmap", case, "cell ids differ between slice counts (S=%d)" % S_)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", default="gemm,attention,layernorm,aggregate,agg_inc,nav,linear_bwd,attn_bwd,attn_train,agg_bwd,gridmap")
    a = ap.parse_args()
    rs = np.random.RandomState(a.seed)
    table = {"gemm": fuzz_gemm, "attention": fuzz_attention, "layernorm": fuzz_layernorm, "aggregate": fuzz_aggregate,
             "nav": lambda r, n: fuzz_nav(r, max(4, n // 4)), "linear_bwd": fuzz_linear_bwd,
             "attn_bwd": lambda r, n: fuzz_attention_bwd(r, max(4, n // 2)),
             "attn_train": lambda r, n: fuzz_attention_train(r, max(4, n // 2)),
             "agg_bwd": lambda r, n: fuzz_aggregate_bwd(r, max(4, n // 4)),
             "gridmap": lambda r, n: fuzz_gridmap(r, max(4, n // 4)),
             "agg_inc": lambda r, n: fuzz_aggregate_incremental(r, max(4, n // 2))}
    for name in a.only.split(","):
        before = len(FAILS)
        table[name](rs, a.cases)
        print("%-10s done, %d failures" % (name, len(FAILS) - before), flush=True)
    return 1 if FAILS else 0


if __name__ == "__main__":
    sys.exit(main())
