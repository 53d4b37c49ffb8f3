"""Differentiable ('grad enabled') forward of GlocalTextPathNavCMT over gridmm_amd.autograd.

Same three modes, same outputs as gridmm_amd/vilmodel.py's inference path, but every Linear / LayerNorm /
attention / GELU / aggregation records a torch.autograd node whose backward is a HIP kernel.  This is what
the fine-tune loop (map_nav_src/r2r/agent_base.py:164-211 -> agent.py:268-451) and the pre-training loop
back-propagate through.  torch here only concatenates, gathers (embedding tables), adds and masks.

Reference line numbers are map_nav_src/models/vilmodel.py unless stated.

Dropout (module.training only): hidden-state dropout (BertSelfOutput / BertOutput / embeddings, the pre-LN layers'
dropout1/2/dropout; ClsPrediction has none) through torch's RNG; dropout on the attention PROBABILITIES
(attention_probs_dropout_prob, vilmodel.py:112,143,334,362; nn.MultiheadAttention(dropout=p), transformer.py:138) inside
the fused attention kernels from a counter-based hash, so the backward regenerates the forward's mask.
"""
import os

import torch
import torch.nn.functional as F

from . import autograd as ag, hostsync as hs, ops
from .grid_memory import pack_reference_lists

N_CELLS = 196


# hostsync.boundary levels (order in which the backward reaches them): 1 = the map / view-point streams entering the local
# encoder, 2 = the outputs of the text and panorama encoders, 3.. = inside the text encoder
CUT_MAP, CUT_ENC, CUT_TEXT = 1, 2, 2


def _drop(model, x, p=None):
    p = model.config.hidden_dropout_prob if p is None else p
    if not (model.training and p > 0):
        return x
    return ag.dropout(x, p) if (x.is_cuda and x.dtype == torch.float32 and x.numel() % 4 == 0) else F.dropout(x, p, True)


def _hidden_p(model):
    """hidden_dropout_prob in train() mode, for the LayerNorms that apply the dropout of the dense layer in front of them."""
    return float(model.config.hidden_dropout_prob) if model.training else 0.0


def _attn_p(model):
    """attention_probs_dropout_prob (vilmodel.py:112,334), active in train() only."""
    return float(model.config.attention_probs_dropout_prob) if model.training else 0.0


def _cat_linear(x, mods, residual=None, out_planes=False):
    """One GEMM for several Linear modules sharing the input (fused q|k|v projections).  out_planes: the result also carries
    its bf16 planes (what the attention kernels of the differentiable path read)."""
    if len(mods) == 1:
        return ag.linear(x, mods[0].weight, mods[0].bias, residual, out_planes=out_planes)
    return ag.linear_group(x, [m.weight for m in mods], [m.bias for m in mods], residual, out_planes=out_planes)


def self_attention_block(model, att, x, kmask):
    """BertAttention (:172-182): LN(dropout(dense(attn(x))) + x)."""
    s = att.self
    qkv = _cat_linear(x, [s.query, s.key, s.value], out_planes=x.shape[-1])
    ctx = ag.self_attention(qkv, kmask, model.heads, _attn_p(model))
    h = ag.linear(ctx, att.output.dense.weight, att.output.dense.bias)
    return ag.layer_norm(h, att.output.LayerNorm, residual=x, dropout_p=_hidden_p(model))


def cross_attention_block(model, xatt, x, ctx_kv, ctx_mask, kv_col=0):
    """BertXAttention (:370-379); ctx_kv = [k | v] projections of the context (possibly several layers wide)."""
    q = ag.linear(x, xatt.att.query.weight, xatt.att.query.bias, out_planes=True)
    c = ag.cross_attention(q, ctx_kv, ctx_mask, model.heads, kv_col=kv_col, dropout_p=_attn_p(model))
    h = ag.linear(c, xatt.output.dense.weight, xatt.output.dense.bias)
    return ag.layer_norm(h, xatt.output.LayerNorm, residual=x, dropout_p=_hidden_p(model))


def ffn_block(model, inter, out, x):
    """BertIntermediate + BertOutput (:185-211)."""
    h = ag.gelu(ag.linear(x, inter.dense.weight, inter.dense.bias))
    o = ag.linear(h, out.dense.weight, out.dense.bias)
    return ag.layer_norm(o, out.LayerNorm, residual=x, dropout_p=_hidden_p(model))


FUSED_XLAYER = True     # a whole cross-modal layer as ONE autograd node (ag.x_layer_fused: one C call forward, one backward)
FUSED_BERT_LAYER = bool(int(os.environ.get("GRIDMM_FUSED_BERT_LAYER", "1")))   # the same for BertLayer (the text encoder)
FUSED_PRELN_LAYER = bool(int(os.environ.get("GRIDMM_FUSED_PRELN_LAYER", "1")))  # ... and the pre-LN layers (panorama / grid encoders)


def _fusable(model, x, inter):
    return FUSED_XLAYER and x.is_cuda and x.shape[-1] == model.heads * 64 and inter.dense.weight.shape[0] % 32 == 0


def bert_layer(model, layer, x, kmask):
    """BertLayer.forward (:214-231)."""
    if FUSED_BERT_LAYER and x.dim() == 3 and _fusable(model, x, layer.intermediate):
        return ag.bert_layer_fused(x, kmask, model.heads, _hidden_p(model), _attn_p(model), layer.attention,
                                   layer.intermediate, layer.output)
    return ffn_block(model, layer.intermediate, layer.output, self_attention_block(model, layer.attention, x, kmask))


def x_layer(model, layer, ctx_kv, ctx_mask, visn, visn_mask, kv_col=0):
    """GraphLXRTXLayer.forward, graph_sprels=None (:399-414)."""
    if _fusable(model, visn, layer.visn_inter):
        return ag.x_layer_fused(visn, ctx_kv, ctx_mask, visn_mask, kv_col, model.heads, _hidden_p(model), _attn_p(model),
                                layer.visual_attention, layer.visn_self_att, layer.visn_inter, layer.visn_output)
    a = cross_attention_block(model, layer.visual_attention, visn, ctx_kv, ctx_mask, kv_col)
    a = self_attention_block(model, layer.visn_self_att, a, visn_mask)
    return ffn_block(model, layer.visn_inter, layer.visn_output, a)


def lang2visn_layer(model, layer, lang, lang_mask, visn, visn_mask):
    """GraphLXRTXLayer.forward_lang2visn (:416-427): text attends to vision, then text self-attention + FFN."""
    xa = layer.visual_attention
    kv = _cat_linear(visn, [xa.att.key, xa.att.value], out_planes=0)
    if _fusable(model, lang, layer.lang_inter):
        return ag.x_layer_fused(lang, kv, visn_mask, lang_mask, 0, model.heads, _hidden_p(model), _attn_p(model),
                                xa, layer.lang_self_att, layer.lang_inter, layer.lang_output)
    a = cross_attention_block(model, xa, lang, kv, visn_mask)
    a = self_attention_block(model, layer.lang_self_att, a, lang_mask)
    return ffn_block(model, layer.lang_inter, layer.lang_output, a)


def pre_ln_encoder(model, enc, x, kmask):
    """TransformerEncoder with normalize_before=True (transformer.py:170-182) + final LayerNorm."""
    p = model.config.hidden_dropout_prob   # create_transformer_encoder passes it as the layer dropout (ops.py:11-16)
    for layer in enc.layers:
        if FUSED_PRELN_LAYER and x.dim() == 3 and x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == model.heads * 64 \
                and layer.linear1.weight.shape[0] % 32 == 0 and x.shape[1] <= 2048:
            x = ag.pre_ln_layer_fused(x, kmask, model.heads, p if model.training else 0.0, layer)   # one autograd node
            continue
        h = ag.layer_norm(x, layer.norm1)
        qkv = ag.linear(h, layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias, out_planes=h.shape[-1])
        ctx = ag.self_attention(qkv, kmask, model.heads, p if model.training else 0.0)   # nn.MultiheadAttention(dropout=p)
        x = x + _drop(model, ag.linear(ctx, layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias), p)
        h = ag.layer_norm(x, layer.norm2)
        f = _drop(model, ag.gelu(ag.linear(h, layer.linear1.weight, layer.linear1.bias)), p)
        x = x + _drop(model, ag.linear(f, layer.linear2.weight, layer.linear2.bias), p)
    return ag.layer_norm(x, enc.norm)


def cls_head(head, x):
    """ClsPrediction (:663-674): Linear -> ReLU -> LN -> Linear(H,1)."""
    h = ag.relu(ag.linear(x, head.net[0].weight, head.net[0].bias))
    h = ag.layer_norm(h, head.net[2])
    return ag.linear(h, head.net[3].weight, head.net[3].bias).squeeze(-1)


# ------------------------------------------------------------------------------------------------
# modes
# ------------------------------------------------------------------------------------------------
def text_embeddings(model, txt_ids):
    """BertEmbeddings (:60-80)."""
    e = model.embeddings
    L = txt_ids.shape[1]
    pos = torch.arange(L, device=txt_ids.device).unsqueeze(0).expand_as(txt_ids)
    x = ag.add_row(e.word_embeddings(txt_ids) + e.position_embeddings(pos), e.token_type_embeddings.weight, 0)
    return _drop(model, ag.layer_norm(x, e.LayerNorm))


def forward_text(model, txt_ids, txt_masks):
    """:730-734."""
    x = text_embeddings(model, txt_ids)
    layers = model.lang_encoder.layer
    n = len(layers)
    for i, layer in enumerate(layers):
        x = bert_layer(model, layer, x, txt_masks)
        if i + 1 < n and (n - 1 - i) % 3 == 0:
            # backward segments of the captured multi-rank step (hostsync.boundary; identity otherwise): the text encoder
            # holds a third of the parameters and is the LAST thing backward reaches -- cut every three layers so that its
            # gradient buckets leave while the layers below are still in backward
            x = hs.boundary(CUT_TEXT + (n - 1 - i) // 3, x)
    return x


def interleave_view_obj(view_embeds, obj_embeds, view_lens, obj_lens):
    """pad_tensors_wgrad([cat(view[:vl], obj[:ol])]) (:754-763) as two gathers + a select: token p of panorama b is
    view p if p < vl_b, object p - vl_b if p < vl_b + ol_b, else zero padding.  Data movement only (differentiable)."""
    B, Vv, H = view_embeds.shape
    Vo = obj_embeds.shape[1]
    vl, ol = view_lens.long().unsqueeze(1), obj_lens.long().unsqueeze(1)
    P = hs.host(lambda: int((vl + ol).max()))
    p = torch.arange(P, device=view_embeds.device).unsqueeze(0).expand(B, P)
    from_view, from_obj = p < vl, (p >= vl) & (p < vl + ol)
    vg = view_embeds.gather(1, p.clamp(max=Vv - 1).unsqueeze(-1).expand(B, P, H))
    og = obj_embeds.gather(1, (p - vl).clamp(0, max(Vo - 1, 0)).unsqueeze(-1).expand(B, P, H))
    zero = torch.zeros((), dtype=view_embeds.dtype, device=view_embeds.device)
    return torch.where(from_view.unsqueeze(-1), vg, torch.where(from_obj.unsqueeze(-1), og, zero)).contiguous()


def forward_panorama(model, view_img_fts, obj_img_fts, loc_fts, nav_types, view_lens, obj_lens):
    """:736-780."""
    ie = model.img_embeddings
    x = ag.layer_norm(ag.linear(view_img_fts.float(), ie.img_linear.weight, ie.img_linear.bias), ie.img_layer_norm)
    lens = view_lens
    if obj_img_fts is not None:
        if ie.obj_linear is None:
            o = ag.layer_norm(ag.linear(obj_img_fts.float(), ie.img_linear.weight, ie.img_linear.bias), ie.img_layer_norm)
        else:
            o = ag.layer_norm(ag.linear(obj_img_fts.float(), ie.obj_linear.weight, ie.obj_linear.bias), ie.obj_layer_norm)
        x = interleave_view_obj(x, o, view_lens, obj_lens)
        lens = view_lens + obj_lens
    y = ag.layer_norm(ag.linear(loc_fts.float(), ie.loc_linear.weight, ie.loc_linear.bias), ie.loc_layer_norm)
    x = ag.add_row(x + y + ag.small_embedding(nav_types, ie.nav_type_embedding.weight),
                   model.embeddings.token_type_embeddings.weight, 1)
    x = _drop(model, ag.layer_norm(x, ie.layer_norm))
    # the mask width is max(lens): a host decision -- the batched collator knows it (collate.NavCollator tags view_lens), any
    # other caller costs one read-back
    width = getattr(lens, "_gridmm_host_max", None)
    masks = torch.arange(hs.host(lambda: int(lens.max()) if width is None else int(width)), device=lens.device).unsqueeze(0) < lens.unsqueeze(1)
    if ie.pano_encoder is not None:
        x = pre_ln_encoder(model, ie.pano_encoder, x, masks)
    return x, masks


def grid_cells(model, txt_embeds, grid_fts, grid_map, gridmap_pos_fts, grid_memory=None, proj_weight=None,
               proj_bias=None):
    """:793-823 -> compacted cell tokens (B,196,H) (zeros past each episode's occupied count) and their mask with
    the reference's view quirk (:817-821), both padded to 196."""
    B, L, H = txt_embeds.shape
    dev = txt_embeds.device
    text_fts = ag.linear(txt_embeds, model.text_proj.weight, model.text_proj.bias)
    if grid_memory is not None:
        slab, perm, cell_start = grid_memory.slab, grid_memory.perm, grid_memory.cell_start
        if gridmap_pos_fts is None:
            gridmap_pos_fts = ag.kernel_copy(grid_memory.pos_fts)    # the buffer is overwritten by the next step
    else:
        slab, perm, cell_start = hs.host(lambda: pack_reference_lists(grid_fts, grid_map))
    cells, occ = ag.grid_aggregate(text_fts, slab, perm, cell_start)
    w = model.grid_proj.weight if proj_weight is None else proj_weight
    proj = ag.linear(cells, w, model.grid_proj.bias if proj_bias is None else proj_bias)                 # grid_proj after the reduction (sum a_j = 1)
    gp = model.grid_pos_embeddings
    pos = ag.layer_norm(ag.linear(gridmap_pos_fts.float(), gp[0].weight, gp[0].bias), gp[1])
    # compaction (occupied cells first, in cell order, zeros behind) + the key mask with the reference's stale-ones quirk:
    # the kernel of the inference path, its backward scatters the rows back (csrc/train_rowops.hip)
    x, mask = ag.cells_compact(proj, pos, occ)
    return x, mask.bool()


def fuse_logits(g_raw, l_raw, grid_raw, fuse_raw, gmap_masks, gmap_visited_masks, vp_nav_masks, cand_of_node,
                cand_visited):
    """:859-899 with the integer index maps of GlocalTextPathNavCMT._fusion_index_maps, differentiable: on the GPU the
    library's forward / backward kernels (gridmm_fuse_logits, gridmm_fuse_logits_bwd); the torch expression below is the
    CPU restatement the tests compare them with."""
    if g_raw.is_cuda:
        return ag.fuse_logits(g_raw, l_raw, grid_raw, fuse_raw, gmap_masks, gmap_visited_masks, vp_nav_masks, cand_of_node,
                              cand_visited)
    ninf = -float("inf")
    fw = torch.sigmoid(fuse_raw).unsqueeze(1) if fuse_raw is not None else 0.5
    global_logits = (g_raw * fw).masked_fill(gmap_visited_masks.bool(), ninf).masked_fill(~gmap_masks.bool(), ninf)
    grid_logits = grid_raw.masked_fill(gmap_visited_masks.bool(), ninf).masked_fill(~gmap_masks.bool(), ninf)
    local_logits = (l_raw * (1 - fw)).masked_fill(~vp_nav_masks.bool(), ninf)
    cv = cand_visited.bool()
    bw = torch.where(cv, local_logits, torch.zeros_like(local_logits)).sum(1, keepdim=True)        # (B,1)
    idx = cand_of_node.long()
    picked = torch.where(idx >= 0, local_logits.gather(1, idx.clamp(min=0)), bw.expand(-1, idx.shape[1]))
    add = torch.where(idx >= -1, picked, torch.zeros_like(picked))
    add = torch.cat([local_logits[:, :1], add[:, 1:]], 1)
    return global_logits, local_logits, grid_logits, global_logits + add


def encode_navigation(model, txt_embeds, txt_masks, cells, cell_masks, gmap_img_embeds, gmap_step_ids, gmap_pos_fts,
                      gmap_masks, vp_img_embeds, vp_pos_fts, vp_masks):
    """:826-856 -> (gmap_embeds (B,G,H), vp_embeds (B,V,H), map_embeds (B,196+G,H)); shared with the VLN-CE twin."""
    H = txt_embeds.shape[-1]
    G = gmap_masks.shape[1]
    ge, le = model.global_encoder, model.local_encoder
    gmap_embeds = gmap_img_embeds.float() + ge.gmap_step_embeddings(gmap_step_ids) + ag.layer_norm(
        ag.linear(gmap_pos_fts.float(), ge.gmap_pos_embeddings[0].weight, ge.gmap_pos_embeddings[0].bias),
        ge.gmap_pos_embeddings[1])
    vp_embeds = vp_img_embeds.float() + ag.layer_norm(
        ag.linear(vp_pos_fts.float(), le.vp_pos_embeddings[0].weight, le.vp_pos_embeddings[0].bias),
        le.vp_pos_embeddings[1])

    map_embeds = torch.cat([cells, gmap_embeds], 1)
    map_masks = torch.cat([cell_masks, gmap_masks], 1)
    map_embeds = pre_ln_encoder(model, model.grid_encoder, map_embeds, map_masks)
    for layer in model.grid_txt_encoder.x_layers:
        xa = layer.visual_attention
        kv = _cat_linear(txt_embeds, [xa.att.key, xa.att.value], out_planes=0)
        map_embeds = x_layer(model, layer, kv, txt_masks, map_embeds, map_masks)

    kv_embeds = torch.cat([map_embeds, txt_embeds], 1)
    kv_masks = torch.cat([map_masks, txt_masks], 1)
    q = torch.cat([map_embeds[:, N_CELLS:], vp_embeds], 1)
    q_masks = torch.cat([gmap_masks, vp_masks], 1)
    xl = le.encoder.x_layers
    kv_all = _cat_linear(kv_embeds, [m for l in xl for m in (l.visual_attention.att.key, l.visual_attention.att.value)],
                         out_planes=0)
    # every layer reads ITS [k | v] column block through a split view: the backward of the split is one concatenation of
    # the layers' (B, Sk, 2H) gradients (a shared kv_col form made every layer return a zero-filled full-width tensor,
    # summed three times: 4 fills + 3 adds of the whole 6144-wide context projection per step)
    for layer, kv in zip(xl, ag.split_with_planes(kv_all, 2 * H)):
        q = x_layer(model, layer, kv, kv_masks, q, q_masks)
    return q[:, :G], q[:, G:], map_embeds


def forward_navigation(model, txt_embeds, txt_masks, gmap_img_embeds, gmap_step_ids, gmap_pos_fts, gmap_masks,
                       gmap_pair_dists, gmap_visited_masks, gmap_vpids, vp_img_embeds, vp_pos_fts, vp_masks,
                       vp_nav_masks, vp_obj_masks, vp_cand_vpids, grid_fts, grid_map, gridmap_pos_fts,
                       grid_memory=None, fusion_maps=None):
    """:782-918."""
    dev = txt_embeds.device
    G, V = gmap_masks.shape[1], vp_masks.shape[1]
    txt_embeds = txt_embeds.float()
    txt_masks, gmap_masks, vp_masks = txt_masks.bool(), gmap_masks.bool(), vp_masks.bool()
    cells, cell_masks = grid_cells(model, txt_embeds, grid_fts, grid_map, gridmap_pos_fts, grid_memory)
    gmap_out, vp_out, map_embeds = encode_navigation(model, txt_embeds, txt_masks, cells, cell_masks, gmap_img_embeds,
                                                     gmap_step_ids, gmap_pos_fts, gmap_masks, vp_img_embeds, vp_pos_fts,
                                                     vp_masks)

    fuse_raw = None
    if model.sap_fuse_linear is not None:
        fuse_raw = cls_head(model.sap_fuse_linear, torch.cat([gmap_out[:, 0], vp_out[:, 0]], 1))
    g_raw = cls_head(model.global_sap_head, gmap_out)
    grid_raw = cls_head(model.grid_sap_head, map_embeds[:, N_CELLS:])
    l_raw = cls_head(model.local_sap_head, vp_out)
    if fusion_maps is None:
        a, b = model._fusion_index_maps(gmap_vpids, gmap_visited_masks, vp_cand_vpids, G, V)
        fusion_maps = (a.to(dev), b.to(dev))
    global_logits, local_logits, grid_logits, fused_logits = fuse_logits(
        g_raw, l_raw, grid_raw, fuse_raw, gmap_masks, gmap_visited_masks, vp_nav_masks, fusion_maps[0], fusion_maps[1])
    obj_logits = None
    if vp_obj_masks is not None:
        obj_logits = cls_head(model.og_head, vp_out).masked_fill(vp_obj_masks.logical_not(), -float("inf"))
    return {"gmap_embeds": gmap_out, "vp_embeds": vp_out, "global_logits": global_logits,
            "local_logits": local_logits, "fused_logits": fused_logits, "obj_logits": obj_logits,
            "grid_logits": grid_logits}
