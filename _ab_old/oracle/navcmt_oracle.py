"""CPU oracle for the GridMM navigation forward path.  TEST INFRASTRUCTURE.

A plain-PyTorch (fp32, CPU or any torch device) op-for-op restatement of the
reference model path, written functionally over a `state_dict` so it shares no
code with the product modules in gridmm_amd/.  Only tests/,
__graft_entry__.smoke() and bench.py's baseline legs may import this.

Restates (file:line relative to /root/reference/map_nav_src/models):
  forward_navigation_per_step   vilmodel.py:782-918   (incl. the 196-cell loop :801-807,
                                the in-place mask compaction quirk :817-823 and the
                                fused-logit loops :881-899)
  forward_text                  vilmodel.py:730-734
  forward_panorama_per_step     vilmodel.py:736-780
  BertSelfAttention/BertOutAttention/BertXAttention  vilmodel.py:95-157, 317-379
  GraphLXRTXLayer / CrossmodalEncoder               vilmodel.py:381-414, 451-468
  ClsPrediction                                     vilmodel.py:663-674
  TransformerEncoder(Layer).forward_pre             transformer.py:62-89, 170-182
  extend_neg_masks / gen_seq_masks                  ops.py:25-44

Parity pin: tests/golden/nav_*.npz hold inputs + outputs produced by importing the
reference itself with deterministic weights (oracle/gen_golden.py); this file is
checked against them in tests/test_oracle_navcmt.py (<=2e-5 abs on logits).

Generalisation beyond the reference: the slab feature dim D_in is read from
`text_proj.weight` (D_in, 768) / `grid_proj.weight` (768, D_in); the reference
hard-codes 768 (vilmodel.py:702-703, 789).
"""
import math

import torch
import torch.nn.functional as F

NUM_HEADS = 12
LN_EPS = 1e-12
N_CELLS = 14 * 14


def linear(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def layer_norm(sd, p, x, eps=LN_EPS):
    w = sd[p + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, sd[p + ".bias"], eps)


def gelu_erf(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def extend_neg_masks(masks):
    return (1.0 - masks.unsqueeze(1).unsqueeze(2).to(torch.float)) * -10000.0


def gen_seq_masks(seq_lens, max_len=None):
    if max_len is None:
        max_len = int(max(seq_lens))
    ar = torch.arange(max_len, device=seq_lens.device).unsqueeze(0)
    return ar < seq_lens.unsqueeze(1)


def _heads(x, nh):
    b, l, h = x.shape
    return x.view(b, l, nh, h // nh).permute(0, 2, 1, 3)


def bert_attention_core(sd, p, hidden, context, ext_mask, nh=NUM_HEADS):
    """query/key/value + scaled softmax(QK^T/sqrt(dh) + mask) V   (vilmodel.py:119-157, 343-368)."""
    q = _heads(linear(sd, p + ".query", hidden), nh)
    k = _heads(linear(sd, p + ".key", context), nh)
    v = _heads(linear(sd, p + ".value", context), nh)
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    if ext_mask is not None:
        scores = scores + ext_mask
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).contiguous()
    return ctx.view(ctx.shape[0], ctx.shape[1], -1)


def bert_self_output(sd, p, hidden, inp):
    return layer_norm(sd, p + ".LayerNorm", linear(sd, p + ".dense", hidden) + inp)


def bert_attention(sd, p, x, ext_mask):
    return bert_self_output(sd, p + ".output", bert_attention_core(sd, p + ".self", x, x, ext_mask), x)


def bert_x_attention(sd, p, x, ctx, ext_mask):
    return bert_self_output(sd, p + ".output", bert_attention_core(sd, p + ".att", x, ctx, ext_mask), x)


def bert_ffn(sd, p_inter, p_out, x):
    h = gelu_erf(linear(sd, p_inter + ".dense", x))
    return layer_norm(sd, p_out + ".LayerNorm", linear(sd, p_out + ".dense", h) + x)


def bert_layer(sd, p, x, ext_mask):
    a = bert_attention(sd, p + ".attention", x, ext_mask)
    return bert_ffn(sd, p + ".intermediate", p + ".output", a)


def graph_lxrt_x_layer(sd, p, lang, lang_ext, visn, visn_ext):
    """GraphLXRTXLayer.forward, graph_sprels=None (vilmodel.py:399-414)."""
    a = bert_x_attention(sd, p + ".visual_attention", visn, lang, lang_ext)
    a = bert_attention(sd, p + ".visn_self_att", a, visn_ext)
    return bert_ffn(sd, p + ".visn_inter", p + ".visn_output", a)


def crossmodal_encoder(sd, p, txt, txt_masks, img, img_masks):
    n_layers = 1 + max(int(k[len(p) + 10:].split(".")[0]) for k in sd if k.startswith(p + ".x_layers."))
    txt_ext, img_ext = extend_neg_masks(txt_masks), extend_neg_masks(img_masks)
    for i in range(n_layers):
        img = graph_lxrt_x_layer(sd, "%s.x_layers.%d" % (p, i), txt, txt_ext, img, img_ext)
    return img


def mha_self(sd, p, x, key_padding_mask, nh=NUM_HEADS):
    """nn.MultiheadAttention self-attention, batch-first restatement, -inf key padding."""
    w, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q, k, v = F.linear(x, w, b).chunk(3, dim=-1)
    q, k, v = _heads(q, nh), _heads(k, nh), _heads(v, nh)
    q = q * (1.0 / math.sqrt(q.shape[-1]))
    scores = torch.matmul(q, k.transpose(-1, -2))
    if key_padding_mask is not None:
        scores = scores.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).contiguous()
    ctx = ctx.view(ctx.shape[0], ctx.shape[1], -1)
    return linear(sd, p + ".out_proj", ctx)


def pre_ln_encoder(sd, p, x, key_padding_mask):
    """TransformerEncoder(normalize_before=True, norm=LN(1e-12)), F.gelu (ops.py:11-23)."""
    n_layers = 1 + max(int(k[len(p) + 8:].split(".")[0]) for k in sd if k.startswith(p + ".layers."))
    for i in range(n_layers):
        q = "%s.layers.%d" % (p, i)
        h = layer_norm(sd, q + ".norm1", x, eps=1e-5)
        x = x + mha_self(sd, q + ".self_attn", h, key_padding_mask)
        h = layer_norm(sd, q + ".norm2", x, eps=1e-5)
        x = x + linear(sd, q + ".linear2", F.gelu(linear(sd, q + ".linear1", h)))
    return layer_norm(sd, p + ".norm", x)


def cls_prediction(sd, p, x):
    h = F.relu(linear(sd, p + ".net.0", x))
    return linear(sd, p + ".net.3", layer_norm(sd, p + ".net.2", h))


# --------------------------------------------------------------------------- modes
def forward_text(sd, txt_ids, txt_masks):
    """vilmodel.py:730-734 + BertEmbeddings :73-93 + LanguageEncoder :441-449."""
    L = txt_ids.shape[1]
    pos = torch.arange(L, device=txt_ids.device).unsqueeze(0).expand_as(txt_ids)
    e = (F.embedding(txt_ids, sd["embeddings.word_embeddings.weight"])
         + F.embedding(pos, sd["embeddings.position_embeddings.weight"])
         + F.embedding(torch.zeros_like(txt_ids), sd["embeddings.token_type_embeddings.weight"]))
    e = layer_norm(sd, "embeddings.LayerNorm", e)
    ext = extend_neg_masks(txt_masks)
    n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("lang_encoder.layer."))
    for i in range(n_layers):
        e = bert_layer(sd, "lang_encoder.layer.%d" % i, e, ext)
    return e


def forward_panorama(sd, view_img_fts, loc_fts, nav_types, view_lens):
    """vilmodel.py:736-780, no-object branch."""
    p = "img_embeddings"
    x = layer_norm(sd, p + ".img_layer_norm", linear(sd, p + ".img_linear", view_img_fts))
    x = (x + layer_norm(sd, p + ".loc_layer_norm", linear(sd, p + ".loc_linear", loc_fts))
         + F.embedding(nav_types, sd[p + ".nav_type_embedding.weight"])
         + sd["embeddings.token_type_embeddings.weight"][1].view(1, 1, -1))
    x = layer_norm(sd, p + ".layer_norm", x)
    masks = gen_seq_masks(view_lens)
    if any(k.startswith(p + ".pano_encoder.") for k in sd):
        x = pre_ln_encoder(sd, p + ".pano_encoder", x, masks.logical_not())
    return x, masks


def grid_aggregate(sd, txt_embeds, grid_fts, grid_map, gridmap_pos_fts):
    """vilmodel.py:788-823.  Returns grid_map_embeds (B,Cmax,H), grid_masks (B,Cmax) bool,
    plus the un-compacted (B,196,H) cell sums and the (B,196) 0/1 occupancy for kernel tests."""
    B = len(grid_fts)
    dev = grid_fts[0].device
    H = sd["grid_proj.weight"].shape[0]
    cells = torch.zeros(B, N_CELLS, H, device=dev)
    text_fts = linear(sd, "text_proj", txt_embeds).permute(0, 2, 1)
    occ = [[] for _ in range(B)]
    max_cell_num = 0
    for b in range(B):
        x = grid_fts[b].to(torch.float32)
        w, _ = (x @ text_fts[b]).max(dim=-1)
        x = linear(sd, "grid_proj", x)
        for i in range(N_CELLS):
            sel = grid_map[b] == i
            cf = x[sel]
            occ[b].append(0 if cf.shape[0] == 0 else 1)
            cells[b, i] = (cf * torch.softmax(w[sel], dim=-1).unsqueeze(-1)).sum(-2)
        max_cell_num = max(max_cell_num, sum(occ[b]))
    occ = torch.tensor(occ, device=dev)
    raw_cells, raw_occ = cells.clone(), occ.clone()

    masks = occ.clone()
    embeds = torch.zeros(B, max_cell_num, H, device=dev)
    cells = cells + layer_norm(sd, "grid_pos_embeddings.1", linear(sd, "grid_pos_embeddings.0", gridmap_pos_fts))
    for b in range(B):
        m = masks[b]  # a VIEW: the next two writes change m.sum() (vilmodel.py:817-821)
        embeds[b, :m.sum()] = cells[b][m == 1]
        masks[b, :m.sum()] = 1
        masks[b, m.sum():] = 0
    return embeds, masks[:, :max_cell_num].bool(), raw_cells, raw_occ


def forward_navigation(sd, batch):
    """vilmodel.py:782-918.  `batch` has the keys of vilmodel.py:934-938."""
    txt_embeds, txt_masks = batch["txt_embeds"], batch["txt_masks"]
    gmap_masks, gmap_visited_masks = batch["gmap_masks"], batch["gmap_visited_masks"]
    gmap_vpids, vp_cand_vpids = batch["gmap_vpids"], batch["vp_cand_vpids"]
    vp_masks, vp_nav_masks = batch["vp_masks"], batch["vp_nav_masks"]
    B = len(batch["grid_fts"])

    grid_embeds, grid_masks, _, _ = grid_aggregate(
        sd, txt_embeds, batch["grid_fts"], batch["grid_map"], batch["gridmap_pos_fts"])
    C = grid_embeds.shape[1]

    gmap = (batch["gmap_img_embeds"]
            + F.embedding(batch["gmap_step_ids"], sd["global_encoder.gmap_step_embeddings.weight"])
            + layer_norm(sd, "global_encoder.gmap_pos_embeddings.1",
                         linear(sd, "global_encoder.gmap_pos_embeddings.0", batch["gmap_pos_fts"])))
    vp = batch["vp_img_embeds"] + layer_norm(
        sd, "local_encoder.vp_pos_embeddings.1",
        linear(sd, "local_encoder.vp_pos_embeddings.0", batch["vp_pos_fts"]))

    map_embeds = torch.cat([grid_embeds, gmap], 1)
    map_masks = torch.cat([grid_masks, gmap_masks], 1)
    map_embeds = pre_ln_encoder(sd, "grid_encoder", map_embeds, map_masks.logical_not())
    map_embeds = crossmodal_encoder(sd, "grid_txt_encoder", txt_embeds, txt_masks, map_embeds, map_masks)
    gmap = map_embeds[:, C:]

    kv_masks = torch.cat([map_masks, txt_masks], 1)
    kv = torch.cat([map_embeds, txt_embeds], 1)
    q_masks = torch.cat([gmap_masks, vp_masks], 1)
    q = torch.cat([gmap, vp], 1)
    q = crossmodal_encoder(sd, "local_encoder.encoder", kv, kv_masks, q, q_masks)
    G = gmap_masks.shape[1]
    gmap, vp = q[:, :G], q[:, G:]

    if "sap_fuse_linear.net.0.weight" in sd:
        fw = torch.sigmoid(cls_prediction(sd, "sap_fuse_linear", torch.cat([gmap[:, 0], vp[:, 0]], 1)))
    else:
        fw = 0.5
    global_logits = cls_prediction(sd, "global_sap_head", gmap).squeeze(2) * fw
    global_logits = global_logits.masked_fill(gmap_visited_masks, -float("inf"))
    global_logits = global_logits.masked_fill(gmap_masks.logical_not(), -float("inf"))
    grid_logits = cls_prediction(sd, "grid_sap_head", map_embeds[:, C:]).squeeze(2)
    grid_logits = grid_logits.masked_fill(gmap_visited_masks, -float("inf"))
    grid_logits = grid_logits.masked_fill(gmap_masks.logical_not(), -float("inf"))
    local_logits = cls_prediction(sd, "local_sap_head", vp).squeeze(2) * (1 - fw)
    local_logits = local_logits.masked_fill(vp_nav_masks.logical_not(), -float("inf"))

    fused = global_logits.clone()
    fused[:, 0] += local_logits[:, 0]
    for i in range(B):
        visited = set(v for v, m in zip(gmap_vpids[i], gmap_visited_masks[i]) if m)
        tmp, bw = {}, 0
        for j, cv in enumerate(vp_cand_vpids[i]):
            if j > 0:
                if cv in visited:
                    bw = bw + local_logits[i, j]
                else:
                    tmp[cv] = local_logits[i, j]
        for j, v in enumerate(gmap_vpids[i]):
            if j > 0 and v not in visited:
                fused[i, j] += tmp[v] if v in tmp else bw

    obj_logits = None
    if batch.get("vp_obj_masks") is not None:
        obj_logits = cls_prediction(sd, "og_head", vp).squeeze(2)
        obj_logits = obj_logits.masked_fill(batch["vp_obj_masks"].logical_not(), -float("inf"))
    return {
        "gmap_embeds": gmap, "vp_embeds": vp, "global_logits": global_logits,
        "local_logits": local_logits, "fused_logits": fused, "obj_logits": obj_logits,
        "grid_logits": grid_logits,
    }
