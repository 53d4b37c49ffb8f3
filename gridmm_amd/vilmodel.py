"""GlocalTextPathNavCMT on hand-written HIP kernels -- the drop-in for the reference's model path.

Keeps the reference's public surface (relative to /root/reference/map_nav_src/models):
  GlocalTextPathNavCMT.forward(mode, batch)        vilmodel.py:920-939   modes 'language' | 'panorama' | 'navigation'
  module tree / state_dict keys                     vilmodel.py:676-710   (reference checkpoints load unchanged)
  forward_navigation_per_step                       vilmodel.py:782-918   -> HIP: aggregation, encoders, logit fusion
The nn.Module tree below only HOLDS parameters under the reference's names; all arithmetic of the
three modes runs in libgridmm_hip.so through gridmm_amd.ops.  There is no PyTorch fallback: on a
box without the library or without a GPU, forward() raises.

Additions over the reference API (optional, used by bench / the agent loop):
  batch['grid_memory'] = GridMemoryBatch   device-resident slab + per-cell point lists instead of the
                                           python lists grid_fts / grid_map (which are still accepted)
  slab feature dim D_in != 768             text_proj: Linear(768, D_in), grid_proj: Linear(D_in, 768)
"""
import math
from types import SimpleNamespace

import os

import numpy as np
import torch
from torch import nn

from . import ops, vilmodel_train
from .grid_memory import GridMemoryBatch, pack_reference_lists

N_CELLS = 196


def default_config(**over):
    """bert-base defaults + map_nav_src/models/vlnbert_init.py:38-56."""
    cfg = dict(
        hidden_size=768, num_attention_heads=12, intermediate_size=3072, vocab_size=30522,
        max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu",
        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
        max_action_steps=100, image_feat_size=768, angle_feat_size=4, obj_feat_size=0, obj_loc_size=3,
        num_l_layers=9, num_pano_layers=2, num_x_layers=4, graph_sprels=True, glocal_fuse=True,
        fix_lang_embedding=False, fix_pano_embedding=False, fix_local_branch=False, update_lang_bert=True,
        output_attentions=True, pred_head_dropout_prob=0.1, use_lang2visn_attn=False,
        grid_feat_size=768,  # D_in of the slab (reference: hard-coded 768, vilmodel.py:702-703)
    )
    cfg.update(over)
    return SimpleNamespace(**cfg)


def _get(cfg, name, default=None):
    return getattr(cfg, name, default)


# ------------------------------------------------------------------------------------------------
# parameter containers (names == reference state_dict keys)
# ------------------------------------------------------------------------------------------------
class BertSelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.query = nn.Linear(c.hidden_size, c.hidden_size)
        self.key = nn.Linear(c.hidden_size, c.hidden_size)
        self.value = nn.Linear(c.hidden_size, c.hidden_size)


class BertSelfOutput(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class BertAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self = BertSelfAttention(c)
        self.output = BertSelfOutput(c)


class BertIntermediate(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.intermediate_size)


class BertOutput(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.intermediate_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class BertLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attention = BertAttention(c)
        self.intermediate = BertIntermediate(c)
        self.output = BertOutput(c)


class BertXAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.att = BertSelfAttention(c)  # BertOutAttention has the same parameters (query/key/value)
        self.output = BertSelfOutput(c)


class GraphLXRTXLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        if _get(c, "use_lang2visn_attn", False):
            self.lang_self_att = BertAttention(c)
            self.lang_inter = BertIntermediate(c)
            self.lang_output = BertOutput(c)
        self.visn_self_att = BertAttention(c)
        self.visn_inter = BertIntermediate(c)
        self.visn_output = BertOutput(c)
        self.visual_attention = BertXAttention(c)


class CrossmodalEncoder(nn.Module):
    def __init__(self, c, num_layers):
        super().__init__()
        self.x_layers = nn.ModuleList([GraphLXRTXLayer(c) for _ in range(num_layers)])


class MultiheadAttentionParams(nn.Module):
    """Same parameter names as nn.MultiheadAttention (in_proj_weight/in_proj_bias/out_proj.*)."""

    def __init__(self, h):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * h, h))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * h))
        self.out_proj = nn.Linear(h, h)
        nn.init.xavier_uniform_(self.in_proj_weight)


class PreLNLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self_attn = MultiheadAttentionParams(c.hidden_size)
        self.linear1 = nn.Linear(c.hidden_size, c.intermediate_size)
        self.linear2 = nn.Linear(c.intermediate_size, c.hidden_size)
        self.norm1 = nn.LayerNorm(c.hidden_size)  # eps 1e-5 (transformer.py:146-147)
        self.norm2 = nn.LayerNorm(c.hidden_size)


class PreLNEncoder(nn.Module):
    """create_transformer_encoder(config, n, norm=True)  (ops.py:11-23)."""

    def __init__(self, c, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([PreLNLayer(c) for _ in range(num_layers)])
        self.norm = nn.LayerNorm(c.hidden_size, eps=1e-12)


class BertEmbeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class LanguageEncoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(c) for _ in range(c.num_l_layers)])


class ImageEmbeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.img_linear = nn.Linear(c.image_feat_size, c.hidden_size)
        self.img_layer_norm = nn.LayerNorm(c.hidden_size, eps=1e-12)
        self.loc_linear = nn.Linear(c.angle_feat_size + 3, c.hidden_size)
        self.loc_layer_norm = nn.LayerNorm(c.hidden_size, eps=1e-12)
        if c.obj_feat_size > 0 and c.obj_feat_size != c.image_feat_size:
            self.obj_linear = nn.Linear(c.obj_feat_size, c.hidden_size)
            self.obj_layer_norm = nn.LayerNorm(c.hidden_size, eps=1e-12)
        else:
            self.obj_linear = self.obj_layer_norm = None
        self.nav_type_embedding = nn.Embedding(3, c.hidden_size)
        self.layer_norm = nn.LayerNorm(c.hidden_size, eps=1e-12)
        self.pano_encoder = PreLNEncoder(c, c.num_pano_layers) if c.num_pano_layers > 0 else None


class LocalVPEncoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.vp_pos_embeddings = nn.Sequential(
            nn.Linear(c.angle_feat_size * 2 + 6, c.hidden_size), nn.LayerNorm(c.hidden_size, eps=1e-12))
        self.encoder = CrossmodalEncoder(c, c.num_x_layers)


class GlobalMapEncoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.gmap_pos_embeddings = nn.Sequential(
            nn.Linear(c.angle_feat_size + 3, c.hidden_size), nn.LayerNorm(c.hidden_size, eps=1e-12))
        self.gmap_step_embeddings = nn.Embedding(c.max_action_steps, c.hidden_size)
        self.sprel_linear = nn.Linear(1, 1) if c.graph_sprels else None  # unused on this path (vilmodel.py:577-590)


class ClsPrediction(nn.Module):
    def __init__(self, hidden_size, input_size=None):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(input_size or hidden_size, hidden_size), nn.ReLU(),
                                 nn.LayerNorm(hidden_size, eps=1e-12), nn.Linear(hidden_size, 1))


# ------------------------------------------------------------------------------------------------
# the model
# ------------------------------------------------------------------------------------------------
class GlocalTextPathNavCMT(nn.Module):
    def __init__(self, config=None):
        super().__init__()
        c = self.config = config if config is not None else default_config()
        H = c.hidden_size
        self.embeddings = BertEmbeddings(c)
        self.lang_encoder = LanguageEncoder(c)
        self.img_embeddings = ImageEmbeddings(c)
        self.local_encoder = LocalVPEncoder(c)
        self.global_encoder = GlobalMapEncoder(c)
        self.global_sap_head = ClsPrediction(H)
        self.local_sap_head = ClsPrediction(H)
        self.grid_sap_head = ClsPrediction(H)
        self.grid_encoder = PreLNEncoder(c, 1)
        self.grid_txt_encoder = CrossmodalEncoder(c, 1)  # num_x_layers forced to 1 (vilmodel.py:694)
        self.grid_pos_embeddings = nn.Sequential(nn.Linear(5, H), nn.LayerNorm(H, eps=1e-12))
        d_in = _get(c, "grid_feat_size", 768)
        self.text_proj = nn.Linear(H, d_in)
        self.grid_proj = nn.Linear(d_in, H)
        self.sap_fuse_linear = ClsPrediction(H, input_size=H * 2) if c.glocal_fuse else None
        if c.obj_feat_size > 0:
            self.og_head = ClsPrediction(H)
        self.heads = c.num_attention_heads
        self.differentiable = None
        self._packed = {}
        for m in self.modules():  # BERT-style init (BertPreTrainedModel.init_weights)
            if isinstance(m, (nn.Linear, nn.Embedding)):
                nn.init.normal_(m.weight, std=0.02)
                if isinstance(m, nn.Linear) and m.bias is not None:
                    nn.init.zeros_(m.bias)

    def _differentiable(self):
        """Which forward runs: the autograd-recording one (vilmodel_train.py) when gradients are enabled and the
        module is in train() mode -- or when `self.differentiable = True` forces it (gradient tests in eval mode);
        otherwise the inference path (bf16x3 attention, fused epilogues, hipGraph-capturable)."""
        if not torch.is_grad_enabled():
            return False
        return self.training if self.differentiable is None else bool(self.differentiable)

    # ---- packed (bf16 hi/lo) weights, rebuilt when a parameter changes -------------------------
    def _pack(self, key, weights, biases):
        ver = tuple((w.data_ptr(), w._version) for w in weights) + tuple((b.data_ptr(), b._version) for b in biases)
        ent = self._packed.get(key)
        if ent is None or ent[0] != ver:
            w = weights[0] if len(weights) == 1 else torch.cat([x.detach() for x in weights], 0)
            b = biases[0] if len(biases) == 1 else torch.cat([x.detach() for x in biases], 0)
            ent = (ver, ops.PackedLinear(w, b))
            self._packed[key] = ent
        return ent[1]

    def _lin(self, mod, key):
        return self._pack(key, [mod.weight], [mod.bias])

    def _wt(self, mod, key):
        """[K][H] fp32 image of a small nn.Linear(K, H) weight for the fused embedding kernels, rebuilt on a version change."""
        ver = (mod.weight.data_ptr(), mod.weight._version)
        ent = self._packed.get(key + ".wT")
        if ent is None or ent[0] != ver:
            ent = (ver, ops.linear_wt(mod))
            self._packed[key + ".wT"] = ent
        return ent[1]

    def _qkv(self, att, key, which="qkv"):
        mods = {"qkv": [att.query, att.key, att.value], "kv": [att.key, att.value], "q": [att.query]}[which]
        return self._pack(key + "." + which, [m.weight for m in mods], [m.bias for m in mods])

    # ---- building blocks (activations travel as ops.Act: fp32 and/or bf16 hi/lo planes) ---------
    def _ln(self, mod, x, residual=None, **kw):
        return ops.layernorm(x, mod.weight, mod.bias, mod.eps, residual=residual, **kw)

    def _attend(self, q, k, v, kmask, k2=None, v2=None):
        """q/k/v: ops.Act holding bf16 planes (B,S,n*H) + column offsets (act, col0) -> bf16x3 attention.  k2 / v2: the
        last keys of the context in a second buffer (instruction rows kept per episode)."""
        def sl(t):
            a, c0 = t
            H = self.config.hidden_size
            return a.hi[..., c0:c0 + H], a.lo[..., c0:c0 + H]
        return ops.attention_rows(sl(q), sl(k), sl(v), kmask, heads=self.heads,      # -> planes for the out-proj
                                  k2=None if k2 is None else sl(k2), v2=None if v2 is None else sl(v2))

    def _self_attention(self, att, key, x, kmask):
        """BertAttention (vilmodel.py:172-182): LN(dense(attn(x)) + x).  x: Act(f32 + planes)."""
        H = x.shape[-1]
        qkv = ops.linear(x, self._qkv(att.self, key), want_f32=False, want_planes=True)
        ctx = self._attend((qkv, 0), (qkv, H), (qkv, 2 * H), kmask)
        h = ops.linear(ctx, self._lin(att.output.dense, key + ".o"), residual=x.f32)
        return self._ln(att.output.LayerNorm, h, want_planes=True)

    def _cross_attention(self, xatt, key, x, ctx, ctx_mask, kv=None, kv2=None):
        """BertXAttention (vilmodel.py:370-379).  kv: optional precomputed (Act planes (B,Sk,n*2H), col0); kv2: the same
        for the last rows of the context when they live in a second buffer."""
        H = x.shape[-1]
        q = ops.linear(x, self._qkv(xatt.att, key, "q"), want_f32=False, want_planes=True)
        if kv is None:
            kv = (ops.linear(ctx, self._qkv(xatt.att, key, "kv"), want_f32=False, want_planes=True), 0)
        c = self._attend((q, 0), (kv[0], kv[1]), (kv[0], kv[1] + H), ctx_mask,
                         k2=None if kv2 is None else (kv2[0], kv2[1]), v2=None if kv2 is None else (kv2[0], kv2[1] + H))
        h = ops.linear(c, self._lin(xatt.output.dense, key + ".o"), residual=x.f32)
        return self._ln(xatt.output.LayerNorm, h, want_planes=True)

    def _ffn(self, inter, out, key, x, planes_out=None):
        h = ops.linear(x, self._lin(inter.dense, key + ".i"), act=ops.ACT_GELU, want_f32=False, want_planes=True)
        o = ops.linear(h, self._lin(out.dense, key + ".f"), residual=x.f32)
        return self._ln(out.LayerNorm, o, want_planes=True, planes_out=planes_out)

    def _bert_layer(self, layer, key, x, kmask):
        a = self._self_attention(layer.attention, key + ".att", x, kmask)
        return self._ffn(layer.intermediate, layer.output, key, a)

    def _x_layer(self, layer, key, lang, lang_mask, visn, visn_mask, kv=None, planes_out=None, kv2=None):
        """GraphLXRTXLayer.forward with graph_sprels=None (vilmodel.py:399-414).  One C call per layer
        (gridmm_xattn_layer_fwd) unless per-kernel timing is on (ops.TIMER: the eleven launches are issued one by one)."""
        if ops.TIMER is None and visn.f32 is not None and visn.hi is not None and visn.f32.is_contiguous():
            H = visn.shape[-1]
            if kv is None:
                kv = (ops.linear(lang, self._qkv(layer.visual_attention.att, key + ".x", "kv"), want_f32=False,
                                 want_planes=True), 0)
            sa, ff = layer.visn_self_att, layer
            pws = (self._qkv(layer.visual_attention.att, key + ".x", "q"), self._lin(layer.visual_attention.output.dense, key + ".x.o"),
                   self._qkv(sa.self, key + ".s"), self._lin(sa.output.dense, key + ".s.o"),
                   self._lin(ff.visn_inter.dense, key + ".i"), self._lin(ff.visn_output.dense, key + ".f"))
            lns = (layer.visual_attention.output.LayerNorm, sa.output.LayerNorm, ff.visn_output.LayerNorm)
            sig = pws + tuple(v for ln in lns for p in (ln.weight, ln.bias) for v in (p.data_ptr(), p._version))
            ent = self._packed.get(key + ".xlayer")
            if ent is None or len(ent[0]) != len(sig) or any(a is not b and a != b for a, b in zip(ent[0], sig)):
                ent = (sig, ops.XLayerWeights(*pws, *lns))
                self._packed[key + ".xlayer"] = ent
            return ops.xattn_layer(ent[1], visn, kv[0], kv[1], kv[1] + H, lang_mask, visn_mask, heads=self.heads,
                                   planes_out=planes_out, kv2=None if kv2 is None else (kv2[0], kv2[1], kv2[1] + H))
        a = self._cross_attention(layer.visual_attention, key + ".x", visn, lang, lang_mask, kv=kv, kv2=kv2)
        a = self._self_attention(layer.visn_self_att, key + ".s", a, visn_mask)
        return self._ffn(layer.visn_inter, layer.visn_output, key, a, planes_out=planes_out)

    def _pre_ln_encoder(self, enc, key, x, kmask):
        """TransformerEncoder, normalize_before=True (transformer.py:170-182), final LN eps 1e-12.
        x: fp32 tensor (the residual stream); returns Act(f32 + planes)."""
        H = x.shape[-1]
        for i, layer in enumerate(enc.layers):
            k = "%s.%d" % (key, i)
            h = self._ln(layer.norm1, x, want_f32=False, want_planes=True)
            qkv = ops.linear(h, self._pack(k + ".in", [layer.self_attn.in_proj_weight],
                                           [layer.self_attn.in_proj_bias]), want_f32=False, want_planes=True)
            ctx = self._attend((qkv, 0), (qkv, H), (qkv, 2 * H), kmask)
            x = ops.linear(ctx, self._lin(layer.self_attn.out_proj, k + ".o"), residual=x).f32
            h = self._ln(layer.norm2, x, want_f32=False, want_planes=True)
            f = ops.linear(h, self._lin(layer.linear1, k + ".1"), act=ops.ACT_GELU, want_f32=False, want_planes=True)
            x = ops.linear(f, self._lin(layer.linear2, k + ".2"), residual=x).f32
        return self._ln(enc.norm, x, want_planes=True)

    def _cls(self, head, key, x):
        """ClsPrediction (vilmodel.py:663-674): Linear -> ReLU -> LN -> Linear(H,1)."""
        h = ops.linear(x, self._lin(head.net[0], key), act=ops.ACT_RELU)
        return ops.ln_dot(h, head.net[2].weight, head.net[2].bias, head.net[2].eps, head.net[3].weight.view(-1),
                          head.net[3].bias)

    @staticmethod
    def _u8(m):
        if m.dtype == torch.bool:
            return m.contiguous().view(torch.uint8)      # same bytes, no copy kernel
        return (m if m.dtype == torch.uint8 else m.to(torch.uint8)).contiguous()

    # ---- modes --------------------------------------------------------------------------------
    def forward_text(self, txt_ids, txt_masks):
        """vilmodel.py:730-734.  With grad enabled: the differentiable path (vilmodel_train.py)."""
        if self._differentiable():
            return vilmodel_train.forward_text(self, txt_ids, txt_masks)
        return self._forward_text_infer(txt_ids, txt_masks)

    @torch.no_grad()
    def _forward_text_infer(self, txt_ids, txt_masks):
        e = self.embeddings
        L = txt_ids.shape[1]
        pos = torch.arange(L, device=txt_ids.device).unsqueeze(0).expand_as(txt_ids)
        x = e.word_embeddings.weight[txt_ids]                     # gathers = data movement
        pt = (e.position_embeddings.weight[pos] + e.token_type_embeddings.weight[0]).contiguous()
        x = self._ln(e.LayerNorm, x.contiguous(), residual=pt, want_planes=True)
        m = self._u8(txt_masks)
        for i, layer in enumerate(self.lang_encoder.layer):
            x = self._bert_layer(layer, "lang.%d" % i, x, m)
        return x.f32

    def forward_panorama_per_step(self, view_img_fts, obj_img_fts, loc_fts, nav_types, view_lens, obj_lens):
        """vilmodel.py:736-780 (view-only branch on HIP; objects are concatenated by the caller form)."""
        if self._differentiable():
            return vilmodel_train.forward_panorama(self, view_img_fts, obj_img_fts, loc_fts, nav_types, view_lens,
                                                   obj_lens)
        return self._forward_panorama_infer(view_img_fts, obj_img_fts, loc_fts, nav_types, view_lens, obj_lens)

    @torch.no_grad()
    def _forward_panorama_infer(self, view_img_fts, obj_img_fts, loc_fts, nav_types, view_lens, obj_lens):
        ie = self.img_embeddings
        x = self._ln(ie.img_layer_norm, ops.linear(view_img_fts.float().contiguous(), self._lin(ie.img_linear, "img"))).f32
        lens = view_lens
        if obj_img_fts is not None:     # [views[:view_len] | objects[:obj_len]] per panorama (vilmodel.py:745-764)
            if ie.obj_linear is None:
                o = self._ln(ie.img_layer_norm, ops.linear(obj_img_fts.float().contiguous(), self._lin(ie.img_linear, "img")))
            else:
                o = self._ln(ie.obj_layer_norm, ops.linear(obj_img_fts.float().contiguous(), self._lin(ie.obj_linear, "obj")))
            x = vilmodel_train.interleave_view_obj(x, o.f32, view_lens, obj_lens)
            lens = view_lens + obj_lens
        extra = (ie.nav_type_embedding.weight[nav_types] + self.embeddings.token_type_embeddings.weight[1]).contiguous()
        y = self._ln(ie.loc_layer_norm, ops.linear(loc_fts.float().contiguous(), self._lin(ie.loc_linear, "loc")),
                     add1=extra)
        x = self._ln(ie.layer_norm, x, residual=y.f32).f32
        # (views only: the mask spans the padded view axis as it is -- no read-back of the largest length; a caller may
        # pad the axis beyond it, e.g. to a graph shape bucket.  With objects the interleaved rows end at max(lens).)
        n = x.shape[1] if obj_img_fts is None else int(lens.max())
        masks = torch.arange(n, device=lens.device).unsqueeze(0) < lens.unsqueeze(1)
        if ie.pano_encoder is not None:
            x = self._pre_ln_encoder(ie.pano_encoder, "pano", x, self._u8(masks)).f32
        return x, masks

    @staticmethod
    def _fusion_index_maps(gmap_vpids, gmap_visited_masks, vp_cand_vpids, G, V):
        """Integer form of the vpid-keyed python loops (vilmodel.py:884-899); host side."""
        B = len(gmap_vpids)
        vis = gmap_visited_masks.detach().cpu().numpy() if torch.is_tensor(gmap_visited_masks) else gmap_visited_masks
        vis = np.asarray(vis).tolist()
        # rows as python lists, ONE array conversion at the end: an element-wise store into a torch tensor costs microseconds,
        # and this runs on the host in front of every forward('navigation') that brings vpid lists (2 ms -> 0.1 ms at B = 32)
        con = [[-2] * G for _ in range(B)]
        cvis = [[0] * V for _ in range(B)]
        for i in range(B):
            gv = gmap_vpids[i]
            visited = set(vp for vp, m in zip(gv, vis[i]) if m)
            tmp = {}
            row = cvis[i]
            for j, cv in enumerate(vp_cand_vpids[i]):
                if j > 0:
                    if cv in visited:
                        row[j] = 1
                    else:
                        tmp[cv] = j
            row = con[i]
            for j, vp in enumerate(gv):
                if j > 0 and vp not in visited:
                    row[j] = tmp.get(vp, -1)
        return (torch.from_numpy(np.asarray(con, dtype=np.int32).reshape(B, G)),
                torch.from_numpy(np.asarray(cvis, dtype=np.uint8).reshape(B, V)))

    def fusion_maps(self, batch, device):
        """Device tensors for batch['fusion_maps'] (integer form of the vpid-keyed fusion loops)."""
        G, V = batch["gmap_masks"].shape[1], batch["vp_masks"].shape[1]
        a, b = self._fusion_index_maps(batch["gmap_vpids"], batch["gmap_visited_masks"], batch["vp_cand_vpids"], G, V)
        return a.to(device), b.to(device)

    def forward_navigation_per_step(self, *args, **kwargs):
        """vilmodel.py:782-918 on HIP kernels.  Same arguments, same output dict.  With grad enabled (fine-tune /
        pre-training) the differentiable path of vilmodel_train.py runs; under torch.no_grad() the inference path."""
        if self._differentiable():
            kwargs.pop("instruction_cache", None)      # (inference-only shortcut; the training path records every op)
            return vilmodel_train.forward_navigation(self, *args, **kwargs)
        return self._forward_navigation_infer(*args, **kwargs)

    @torch.no_grad()
    def _forward_navigation_infer(
            self, txt_embeds, txt_masks, gmap_img_embeds, gmap_step_ids, gmap_pos_fts, gmap_masks,
            gmap_pair_dists, gmap_visited_masks, gmap_vpids, vp_img_embeds, vp_pos_fts, vp_masks,
            vp_nav_masks, vp_obj_masks, vp_cand_vpids, grid_fts, grid_map, gridmap_pos_fts, grid_memory=None,
            fusion_maps=None, instruction_cache=None):
        dev = txt_embeds.device
        G, V = gmap_masks.shape[1], vp_masks.shape[1]
        gmap_m = self._u8(gmap_masks)
        gmap_embeds, vp_embeds, map_embeds = self._encode_navigation_infer(
            txt_embeds, txt_masks, gmap_img_embeds, gmap_step_ids, gmap_pos_fts, gmap_masks, vp_img_embeds, vp_pos_fts,
            vp_masks, grid_fts, grid_map, gridmap_pos_fts, grid_memory, icache=instruction_cache)
        return self._heads_infer(gmap_embeds, vp_embeds, map_embeds, gmap_m, gmap_visited_masks, gmap_vpids,
                                 vp_nav_masks, vp_obj_masks, vp_cand_vpids, fusion_maps, G, V, dev)

    # Cell rows of the padded [cells | nodes] sequence.  None: always 196 (no host decision anywhere: the whole step is one
    # hipGraph).  A tuple of ascending bucket sizes ending in 196: the reference's max_cell_num truncation
    # (vilmodel.py:809-823, ops.py:46-68 pad_tensors_wgrad) -- eager calls read the batch's largest occupied-cell count
    # (one small D2H, as the reference's python max() does) and run the encoders on the smallest bucket that holds it;
    # graph.GraphedNavStep keeps one captured back half per bucket and predicts the bucket from the previous step.
    varlen_buckets = None
    DEFAULT_BUCKETS = (64, 80, 96, 112, 128, 144, 160, 176, N_CELLS)   # 16-row steps: a step costs what its occupied cells cost

    # ---- per-episode instruction-side constants of 'navigation' -------------------------------------------------
    # The reference recomputes, at EVERY step, text_proj(txt_embeds) (vilmodel.py:793), the K / V projections of the
    # instruction in the grid / text layer (:841, BertXAttention :370-379) and the K / V projections of the 80
    # instruction rows of the local encoder's [map | txt] context in each of its layers (:846-853) -- from a
    # txt_embeds that the 'language' call produced once per episode (r2r/agent.py:274-276).  Row-wise projections of
    # constant rows: computing them once gives the same bits.  `instruction_cache` does that; 'navigation' takes the
    # result as batch["instruction_cache"] (declared in bench.py's config and DESIGN.md like the grid_proj shortcut).
    def instruction_cache_shapes(self, B, L):
        H, D = self.config.hidden_size, self.text_proj.out_features
        nl = len(self.local_encoder.encoder.x_layers)
        ng = len(self.grid_txt_encoder.x_layers)
        return {"txt_hi": ((B, L, H), torch.bfloat16), "txt_lo": ((B, L, H), torch.bfloat16),
                "frag": ((B, 2, (L + 15) // 16, D // 32, 64, 8), torch.float16),
                "gt_hi": ((ng, B, L, 2 * H), torch.bfloat16), "gt_lo": ((ng, B, L, 2 * H), torch.bfloat16),
                "loc_hi": ((B, L, nl * 2 * H), torch.bfloat16), "loc_lo": ((B, L, nl * 2 * H), torch.bfloat16),
                "txt_m": ((B, L), torch.uint8)}

    @torch.no_grad()
    def instruction_cache(self, txt_embeds, txt_masks, out=None):
        """-> SimpleNamespace of the instruction-side tensors every navigation step of the episode reuses.  out: dict of
        preallocated buffers (instruction_cache_shapes; e.g. the static buffers a hipGraph of the step reads)."""
        B, L, H = txt_embeds.shape
        dev = txt_embeds.device
        if out is None:
            out = {k: torch.empty(shp, dtype=dt, device=dev) for k, (shp, dt) in self.instruction_cache_shapes(B, L).items()}
        txt = ops.split_rows(txt_embeds.float().contiguous(), out=(out["txt_hi"], out["txt_lo"]))
        text_fts = ops.linear(txt, self._lin(self.text_proj, "text_proj")).f32
        frag = ops.text_fragments(text_fts, out=out["frag"])
        gt = []
        for i, layer in enumerate(self.grid_txt_encoder.x_layers):
            ops.linear(txt, self._qkv(layer.visual_attention.att, "grid_txt.%d.x" % i, "kv"), want_f32=False,
                       planes_out=(out["gt_hi"][i], out["gt_lo"][i]))
            gt.append(ops.Act(None, out["gt_hi"][i], out["gt_lo"][i]))
        ops.linear(txt, self._local_kv_pack(), want_f32=False, planes_out=(out["loc_hi"], out["loc_lo"]))
        out["txt_m"].copy_(self._u8(txt_masks))
        return SimpleNamespace(txt=ops.Act(None, out["txt_hi"], out["txt_lo"]), txt_m=out["txt_m"], frag=frag, gt_kv=gt,
                               local_kv=ops.Act(None, out["loc_hi"], out["loc_lo"]), L=L, buffers=out)

    def _local_kv_pack(self):
        xl = self.local_encoder.encoder.x_layers
        return self._pack("local.kv_all",
                          [w for l in xl for w in (l.visual_attention.att.key.weight, l.visual_attention.att.value.weight)],
                          [b for l in xl for b in (l.visual_attention.att.key.bias, l.visual_attention.att.value.bias)])

    @torch.no_grad()
    def _nav_front(self, txt_embeds, txt_masks, grid_fts, grid_map, gridmap_pos_fts, grid_memory=None, txt_planes=None,
                   icache=None):
        """vilmodel.py:793-807: text_proj, instruction-relevance aggregation of the grid memory, grid_proj on the 196
        reduced cell vectors.  Independent of how many cells are occupied.  txt_planes = (hi, lo): where the bf16 planes
        of the instruction go (e.g. the tail of the local encoder's context buffer)."""
        B, L, H = txt_embeds.shape
        txt_src = txt_embeds                                   # the caller's tensor: identity of the instruction side
        if icache is not None:
            txt, frag, txt_m = icache.txt, icache.frag, icache.txt_m
        else:
            txt_embeds = txt_embeds.float().contiguous()
            txt = ops.split_rows(txt_embeds, out=txt_planes)   # fp32 + planes
            text_fts = ops.linear(txt, self._lin(self.text_proj, "text_proj")).f32
            frag = ops.text_fragments(text_fts)
            txt_m = self._u8(txt_masks).contiguous()
        n_points = None
        if grid_memory is not None:
            slab, perm, cell_start = grid_memory.slab, grid_memory.perm, grid_memory.cell_start
            n_points = grid_memory.points_upper_bound()   # the slab is allocated for max_steps observations
            if gridmap_pos_fts is None:
                gridmap_pos_fts = grid_memory.pos_fts
        else:
            slab, perm, cell_start = pack_reference_lists(grid_fts, grid_map)
        res = None
        if (grid_memory is not None and getattr(grid_memory, "relevance_cache_enabled", False)
                and ops.two_pass_aggregation(slab.shape[2], L)
                and (grid_memory.relevance_cache_in_graphs or not torch.cuda.is_current_stream_capturing())):
            # device-resident memory on a two-pass shape: the relevance of the points of earlier steps is kept (it depends on
            # the slab row and the instruction only), this step computes the new observation's and reads the slab once
            tp = self.text_proj
            key = (id(txt_src), txt_src._version, txt_src.data_ptr(), tp.weight.data_ptr(), tp.weight._version,
                   tp.bias._version)
            st = grid_memory.relevance_cache(key, keep_alive=txt_src)       # (holding the tensor keeps its id unique)
            res = ops.grid_aggregate_incremental(slab, perm, cell_start, frag, L, grid_memory.n_pts,
                                                 None if grid_memory._active is None else grid_memory.act_d,
                                                 grid_memory.n_new, st,
                                                 full=bool((grid_memory.n_pts_host <= grid_memory.n_new).all()))
        cells, occ = res if res is not None else ops.grid_aggregate(slab, perm, cell_start, frag, L, n_points=n_points)
        proj = ops.linear(cells, self._lin(self.grid_proj, "grid_proj")).f32
        return SimpleNamespace(txt=txt, txt_m=txt_m, proj=proj, occ=occ, gridmap_pos_fts=gridmap_pos_fts,
                               in_place=txt_planes is not None, icache=icache, L=L)

    @torch.no_grad()
    def _nav_back(self, fr, c_pad, gmap_img_embeds, gmap_step_ids, gmap_pos_fts, gmap_masks, vp_img_embeds, vp_pos_fts,
                  vp_masks, kv=None):
        """vilmodel.py:813-856 on a [cells | nodes] sequence padded to c_pad + G rows (c_pad >= the batch's largest
        occupied-cell count): position embeddings, grid encoder, grid/text layer, local encoder."""
        txt, txt_m = fr.txt, fr.txt_m
        ic = getattr(fr, "icache", None)
        dev = txt.hi.device
        B, L, H = txt.hi.shape
        G, V = gmap_masks.shape[1], vp_masks.shape[1]
        S = c_pad + G
        gmap_m, vp_m = (self._u8(m).contiguous() for m in (gmap_masks, vp_masks))
        if ic is not None:
            kv = ops.Act(None, *ops._planes_like((B, S, H), dev))    # map rows only: the instruction's K / V are cached
        elif kv is None:
            kv = ops.Act(None, *ops._planes_like((B, S + L, H), dev))
            ops.copy_planes(txt, kv, S)                  # (callers that know c_pad up front let _nav_front write in place)
        kv_masks = torch.empty(B, S + L, dtype=torch.uint8, device=dev)
        q_masks = torch.empty(B, G + V, dtype=torch.uint8, device=dev)
        map_masks = kv_masks[:, :S]

        # ---- [cells | gmap nodes] sequence (vilmodel.py:813-837); position embeddings of cells, nodes and views + all
        # byte masks in two launches
        map_embeds = torch.empty(B, S, H, dtype=torch.float32, device=dev)
        gp = self.grid_pos_embeddings
        self._cells = ops.cells_embed(fr.proj, fr.gridmap_pos_fts, gp[0], gp[1], fr.occ, map_embeds, kv_masks,
                                      tail_mask=gmap_m, wT=self._wt(gp[0], "grid_pos"), c_pad=c_pad)
        q = torch.empty(B, G + V, H, dtype=torch.float32, device=dev)
        qp = ops._planes_like((B, G + V, H), dev)
        ge, le = self.global_encoder, self.local_encoder
        ops.node_embed(
            [ops.embed_seg(gmap_pos_fts, ge.gmap_pos_embeddings[0], ge.gmap_pos_embeddings[1], gmap_img_embeds,
                           map_embeds[:, c_pad:], table=ge.gmap_step_embeddings.weight, idx=gmap_step_ids,
                           wT=self._wt(ge.gmap_pos_embeddings[0], "gmap_pos")),
             ops.embed_seg(vp_pos_fts, le.vp_pos_embeddings[0], le.vp_pos_embeddings[1], vp_img_embeds, q[:, G:],
                           planes=(qp[0][:, G:], qp[1][:, G:]), wT=self._wt(le.vp_pos_embeddings[0], "vp_pos"))],
            H, gmap_m, vp_m, txt_m, kv_masks, c_pad, q_masks)

        # ---- grid encoder + grid/text cross-modal layer (vilmodel.py:840-841)
        mp = self._pre_ln_encoder(self.grid_encoder, "grid_enc", map_embeds, map_masks)
        nl = len(self.grid_txt_encoder.x_layers)
        for i, layer in enumerate(self.grid_txt_encoder.x_layers):
            mp = self._x_layer(layer, "grid_txt.%d" % i, txt, txt_m, mp, map_masks,
                               kv=None if ic is None else (ic.gt_kv[i], 0),
                               planes_out=(kv.hi[:, :S], kv.lo[:, :S]) if i == nl - 1 else None)
        map_embeds = mp.f32

        # ---- local encoder over q = [gmap | vp], kv = [map | txt] (vilmodel.py:843-856).  The context is the
        # same for all layers, so the K/V projections of every layer run as ONE GEMM (N = layers * 2H).
        xl = le.encoder.x_layers
        kv_all = ops.linear(kv, self._local_kv_pack(), want_f32=False, want_planes=True)
        ops.copy_rows(map_embeds[:, c_pad:], q, 0)
        ops.copy_planes(ops.Act(None, kv.hi[:, c_pad:S], kv.lo[:, c_pad:S]), ops.Act(None, qp[0], qp[1]), 0)
        qa = ops.Act(q, qp[0], qp[1])
        for i, layer in enumerate(xl):
            qa = self._x_layer(layer, "local.%d" % i, None, kv_masks, qa, q_masks, kv=(kv_all, 2 * H * i),
                               kv2=None if ic is None else (ic.local_kv, 2 * H * i))
        q = qa.f32
        self._last_acts = (qa, kv, kv.hi.shape[1], c_pad)     # planes of the outputs: the heads read them in place
        return q[:, :G], q[:, G:], map_embeds

    @torch.no_grad()
    def navigation_front(self, batch):
        """First half of forward('navigation', batch) (independent of the occupied-cell count): see _nav_front."""
        return self._nav_front(batch["txt_embeds"], batch["txt_masks"], batch.get("grid_fts"), batch.get("grid_map"),
                               batch.get("gridmap_pos_fts"), grid_memory=batch.get("grid_memory"),
                               icache=batch.get("instruction_cache"))

    @torch.no_grad()
    def navigation_back(self, fr, c_pad, batch):
        """Second half on a sequence padded to c_pad cell rows -> the output dict of forward('navigation').  Valid when
        the batch's largest occupied-cell count (self._cells[1], a device int32 written by this call) is <= c_pad."""
        dev = batch["txt_embeds"].device
        G, V = batch["gmap_masks"].shape[1], batch["vp_masks"].shape[1]
        gmap_embeds, vp_embeds, map_embeds = self._nav_back(
            fr, int(c_pad), batch["gmap_img_embeds"], batch["gmap_step_ids"], batch["gmap_pos_fts"], batch["gmap_masks"],
            batch["vp_img_embeds"], batch["vp_pos_fts"], batch["vp_masks"])
        return self._heads_infer(gmap_embeds, vp_embeds, map_embeds, self._u8(batch["gmap_masks"]),
                                 batch["gmap_visited_masks"], batch["gmap_vpids"], batch["vp_nav_masks"],
                                 batch.get("vp_obj_masks"), batch["vp_cand_vpids"], batch.get("fusion_maps"), G, V, dev)

    def pick_bucket(self, cmax):
        """Smallest configured bucket that holds cmax occupied cells (196 when varlen is off)."""
        for c in (self.varlen_buckets or (N_CELLS,)):
            if cmax <= c:
                return int(c)
        return N_CELLS

    @torch.no_grad()
    def _encode_navigation_infer(self, txt_embeds, txt_masks, gmap_img_embeds, gmap_step_ids, gmap_pos_fts, gmap_masks,
                                 vp_img_embeds, vp_pos_fts, vp_masks, grid_fts, grid_map, gridmap_pos_fts,
                                 grid_memory=None, icache=None):
        """vilmodel.py:788-856: aggregation, grid encoder, grid/text layer, local encoder -> (gmap_embeds (B,G,H),
        vp_embeds (B,V,H), map_embeds (B,c_pad+G,H)).  Shared with the VLN-CE twin (gridmap/vilmodel.py:710-776).

        Sequences are never concatenated: ONE plane buffer `kv` (B, c_pad+G+L, H) is the local encoder's [map | txt]
        context (vilmodel.py:846-848); the instruction planes are split straight into its tail, the grid/text layer's
        last LayerNorm writes the map planes into its head, and the GEMMs that need only one part read it in place
        through the batched row map of gridmm_linear_planes_map.  The byte masks live the same way in `kv_masks`."""
        dev = txt_embeds.device
        B, L, H = txt_embeds.shape
        G = gmap_masks.shape[1]
        back = (gmap_img_embeds, gmap_step_ids, gmap_pos_fts, gmap_masks, vp_img_embeds, vp_pos_fts, vp_masks)
        if self.varlen_buckets and not torch.cuda.is_current_stream_capturing():
            fr = self._nav_front(txt_embeds, txt_masks, grid_fts, grid_map, gridmap_pos_fts, grid_memory, icache=icache)
            # the reference's max_cell_num (a host decision): from the grid memory's pinned word when it tracked the
            # count behind its last step (no stall), else read back from the occupancy bytes of this call
            cmax = grid_memory.cmax_hint() if grid_memory is not None and hasattr(grid_memory, "cmax_hint") else None
            if cmax is None:
                cmax = int(fr.occ.sum(1, dtype=torch.int32).max())
            return self._nav_back(fr, self.pick_bucket(cmax), *back)
        if icache is not None:
            fr = self._nav_front(txt_embeds, txt_masks, grid_fts, grid_map, gridmap_pos_fts, grid_memory, icache=icache)
            return self._nav_back(fr, N_CELLS, *back)
        S = N_CELLS + G
        kv = ops.Act(None, *ops._planes_like((B, S + L, H), dev))
        fr = self._nav_front(txt_embeds, txt_masks, grid_fts, grid_map, gridmap_pos_fts, grid_memory,
                             txt_planes=(kv.hi[:, S:], kv.lo[:, S:]))
        return self._nav_back(fr, N_CELLS, *back, kv=kv)

    @torch.no_grad()
    def _heads_infer(self, gmap_embeds, vp_embeds, map_embeds, gmap_m, gmap_visited_masks, gmap_vpids, vp_nav_masks,
                     vp_obj_masks, vp_cand_vpids, fusion_maps, G, V, dev):
        """vilmodel.py:859-907: the ClsPrediction heads as ONE grouped GEMM launch (fuse head = two K-halves over row 0
        of the node / view blocks, global + local (+ object) heads stacked over all G + V rows, grid head over the
        pre-local-encoder node rows read in place from the context planes) + ONE tail / masking / fusion launch."""
        acts, self._last_acts = getattr(self, "_last_acts", None), None
        B, H = gmap_embeds.shape[0], gmap_embeds.shape[-1]
        if fusion_maps is None:   # host-built from the python vpid lists; pass precomputed device tensors to avoid
            cand_of_node, cand_visited = self._fusion_index_maps(gmap_vpids, gmap_visited_masks, vp_cand_vpids, G, V)
            fusion_maps = (cand_of_node.to(dev), cand_visited.to(dev))   # the H2D (needed under graph capture)
        has_obj = vp_obj_masks is not None
        if acts is None or acts[0].f32.data_ptr() != gmap_embeds.data_ptr():
            return self._heads_infer_unfused(gmap_embeds, vp_embeds, map_embeds, gmap_m, gmap_visited_masks, vp_nav_masks,
                                             vp_obj_masks, fusion_maps, G, V)
        qa, kv, SL, c_pad = acts
        Sq = G + V
        gh, lh = self.global_sap_head.net, self.local_sap_head.net
        stack = [gh[0], lh[0]] + ([self.og_head.net[0]] if has_obj else [])
        pw_gl = self._pack("ghead+lhead+og" if has_obj else "ghead+lhead", [m.weight for m in stack], [m.bias for m in stack])
        pw_grid = self._lin(self.grid_sap_head.net[0], "gridhead")
        h_gl = torch.empty(B * Sq, pw_gl.N, dtype=torch.float32, device=dev)
        h_grid = torch.empty(B * G, H, dtype=torch.float32, device=dev)
        probs = [ops.gemm_problem(qa.hi, qa.lo, H, B * Sq, pw_gl, h_gl, act=ops.ACT_RELU),
                 ops.gemm_problem(kv.hi, kv.lo, H, B * G, pw_grid, h_grid, act=ops.ACT_RELU, a_rpb=G, a_bs=SL * H,
                                  a_off=c_pad * H)]
        fa = fb = fbias = None
        if self.sap_fuse_linear is not None:
            pw_f = self._lin(self.sap_fuse_linear.net[0], "fuse")
            fa = torch.empty(B, H, dtype=torch.float32, device=dev)
            fb = torch.empty(B, H, dtype=torch.float32, device=dev)
            fbias = pw_f.bias
            probs += [ops.gemm_problem(qa.hi, qa.lo, Sq * H, B, pw_f, fa, K=H, bias=False),
                      ops.gemm_problem(qa.hi, qa.lo, Sq * H, B, pw_f, fb, K=H, bias=False, a_off=G * H, w_col0=H)]
        ops.linear_grouped(probs)
        tails = [ops.cls_tail(self.sap_fuse_linear.net) if self.sap_fuse_linear is not None else None,
                 ops.cls_tail(gh), ops.cls_tail(lh), ops.cls_tail(self.grid_sap_head.net),
                 ops.cls_tail(self.og_head.net) if has_obj else None]
        global_logits, local_logits, grid_logits, fused_logits, obj_logits = ops.nav_heads(
            h_gl, fa, fb, fbias, h_grid, tails, gmap_m.contiguous(), self._u8(gmap_visited_masks).contiguous(),
            self._u8(vp_nav_masks).contiguous(), self._u8(vp_obj_masks).contiguous() if has_obj else None,
            fusion_maps[0], fusion_maps[1], G, V)
        return {
            "gmap_embeds": gmap_embeds, "vp_embeds": vp_embeds, "global_logits": global_logits,
            "local_logits": local_logits, "fused_logits": fused_logits, "obj_logits": obj_logits,
            "grid_logits": grid_logits,
        }

    @torch.no_grad()
    def _heads_infer_unfused(self, gmap_embeds, vp_embeds, map_embeds, gmap_m, gmap_visited_masks, vp_nav_masks,
                             vp_obj_masks, fusion_maps, G, V):
        """The same heads from plain fp32 embeddings, one launch per op (callers that hand in their own tensors)."""
        fuse_raw = None
        if self.sap_fuse_linear is not None:
            fuse_raw = self._cls(self.sap_fuse_linear, "fuse", torch.cat([gmap_embeds[:, 0], vp_embeds[:, 0]], 1))
        g_raw = self._cls(self.global_sap_head, "ghead", gmap_embeds)
        grid_raw = self._cls(self.grid_sap_head, "gridhead", map_embeds[:, map_embeds.shape[1] - G:])
        l_raw = self._cls(self.local_sap_head, "lhead", vp_embeds)
        global_logits, local_logits, grid_logits, fused_logits = ops.fuse_logits(
            g_raw, l_raw, grid_raw, fuse_raw, gmap_m, self._u8(gmap_visited_masks), self._u8(vp_nav_masks),
            fusion_maps[0], fusion_maps[1])
        obj_logits = None
        if vp_obj_masks is not None:
            obj_logits = self._cls(self.og_head, "oghead", vp_embeds)
            obj_logits.masked_fill_(vp_obj_masks.logical_not(), -float("inf"))
        return {
            "gmap_embeds": gmap_embeds, "vp_embeds": vp_embeds, "global_logits": global_logits,
            "local_logits": local_logits, "fused_logits": fused_logits, "obj_logits": obj_logits,
            "grid_logits": grid_logits,
        }

    def forward(self, mode, batch, **kwargs):
        """vilmodel.py:920-939."""
        if mode == "language":
            return self.forward_text(batch["txt_ids"], batch["txt_masks"])
        elif mode == "panorama":
            return self.forward_panorama_per_step(
                batch["view_img_fts"], batch.get("obj_img_fts"), batch["loc_fts"], batch["nav_types"],
                batch["view_lens"], batch.get("obj_lens"))
        elif mode == "navigation":
            return self.forward_navigation_per_step(
                batch["txt_embeds"], batch["txt_masks"], batch["gmap_img_embeds"], batch["gmap_step_ids"],
                batch["gmap_pos_fts"], batch["gmap_masks"], batch.get("gmap_pair_dists"),
                batch["gmap_visited_masks"], batch["gmap_vpids"], batch["vp_img_embeds"], batch["vp_pos_fts"],
                batch["vp_masks"], batch["vp_nav_masks"], batch.get("vp_obj_masks"), batch["vp_cand_vpids"],
                batch.get("grid_fts"), batch.get("grid_map"), batch.get("gridmap_pos_fts"),
                grid_memory=batch.get("grid_memory"), fusion_maps=batch.get("fusion_maps"),
                **({"instruction_cache": batch["instruction_cache"]} if batch.get("instruction_cache") is not None else {}))
        raise NotImplementedError("wrong mode: %s" % mode)
