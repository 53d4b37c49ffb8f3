"""Pre-training twin: GlocalTextPathCMTPreTraining on the HIP autograd path.

Mirrors (relative to /root/reference/pretrain_src/model):
  GlocalTextPathCMTPreTraining.forward(batch, task, compute_loss)   pretrain_cmt.py:71-129
      forward_mlm :131-153   forward_mrc :161-213   forward_sap :215-290   forward_og :292-321
  GlocalTextPathCMT.forward / forward_mlm                            vilmodel.py:668-766 / :767-856
  ImageEmbeddings.forward :483-532, GlobalMapEncoder._aggregate_gmap_features / gmap_input_embedding :569-620,
  LocalVPEncoder.vp_input_embedding :545-560, BertOnlyMLMHead :262-303
Same module tree / state_dict keys (`bert.*`, `mlm_head.predictions.*`, `image_classifier.net.*`, `*_sap_head.net.*`;
grid_proj stored in fp16 as the reference does, vilmodel.py:664), same batch dict (tasks.py collates), same per-sample
loss vectors.  Every Linear / LayerNorm / attention / GELU / aggregation, forward and backward, is a HIP kernel
(gridmm_amd.autograd); torch gathers, pads and evaluates the scalar losses.

Numerics vs the reference: the reference projects every grid point with an fp16 grid_proj and reduces in fp16
(vilmodel.py:693-703); here the softmax-weighted reduction runs in fp32 and grid_proj (weights up-cast) is applied
to the 196 reduced vectors -- identical algebra, ~1e-3 relative difference from the reference's own fp16 rounding.
"""
from collections import defaultdict

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import autograd as ag, hostsync as hs, vilmodel as V, vilmodel_train as VT

N_CELLS = 196


PAD_VOCAB = bool(int(__import__('os').environ.get('GRIDMM_PAD_VOCAB', '1')))   # A/B switch of forward_mlm's decoder call

class GlocalTextPathCMT(nn.Module):
    """The pre-training backbone (pretrain_src/model/vilmodel.py:640-856): parameters under the reference's names and
    its two entry points forward(...) / forward_mlm(...) with the reference's positional signature; the arithmetic runs
    on the HIP autograd path (vilmodel_train.py)."""

    def __init__(self, c):
        super().__init__()
        self.config = c
        H = c.hidden_size
        self.embeddings = V.BertEmbeddings(c)
        self.lang_encoder = V.LanguageEncoder(c)
        self.img_embeddings = V.ImageEmbeddings(c)
        self.local_encoder = V.LocalVPEncoder(c)                   # num_x_layers as configured ...
        self.global_encoder = V.GlobalMapEncoder(c)
        self.global_encoder.sprel_linear = None                    # vilmodel.py:576
        self.grid_encoder = V.PreLNEncoder(c, 1)
        self.grid_txt_encoder = V.CrossmodalEncoder(c, 1)          # ... but 1 here (vilmodel.py:654-655)
        self.grid_pos_embeddings = nn.Sequential(nn.Linear(5, H), nn.LayerNorm(H, eps=1e-12))
        self.text_proj = nn.Linear(H, H)
        self.grid_proj = nn.Linear(H, H).to(torch.float16)         # vilmodel.py:664
        self.heads = c.num_attention_heads

    # ---- shared encoder front (vilmodel.py:668-738) ------------------------------------------------------------
    def _front(self, batch):
        """-> dict(txt_embeds, txt_masks, map_embeds, map_masks, gmap_masks, vp_input, vp_masks)."""
        b = self
        dev = batch["txt_ids"].device
        txt_masks = _seq_masks(batch["txt_lens"], batch["txt_ids"].shape[1])
        # (hs.boundary: separators of the autograd graph for the segmented backward of the captured multi-rank step --
        # identity unless hostsync.CUTS is active)
        txt_embeds = hs.boundary(VT.CUT_ENC, VT.forward_text(b, batch["txt_ids"].long(), txt_masks))
        cells, cell_masks = VT.grid_cells(b, txt_embeds, batch["grid_fts"], batch["grid_map"],
                                          batch["gridmap_pos_fts"], proj_weight=b.grid_proj.weight.float(),
                                          proj_bias=b.grid_proj.bias.float())

        # trajectory embedding: every step's panorama through the pano encoder (ImageEmbeddings.forward)
        only_view_lens = batch["traj_vp_view_lens"].long()
        obj_lens = batch["traj_vp_obj_lens"].long() if batch["traj_obj_img_fts"] is not None else None
        traj, traj_masks = VT.forward_panorama(b, batch["traj_view_img_fts"], batch["traj_obj_img_fts"],
                                               batch["traj_loc_fts"], batch["traj_nav_types"].long(), only_view_lens,
                                               obj_lens)
        traj = hs.boundary(VT.CUT_ENC, traj)
        step_lens = [int(x) for x in batch["traj_step_lens"]]
        view_lens = only_view_lens if obj_lens is None else only_view_lens + obj_lens      # traj_vp_lens (:512-516)
        Vmax, H = traj.shape[1], traj.shape[2]
        B = len(step_lens)
        offs = [0]
        for t in step_lens:
            offs.append(offs[-1] + t)

        # global-map node features: masked means over panorama tokens, as one (B, G-1, T*Vmax) weight matrix built on
        # the host from the vpid lists (GlobalMapEncoder._aggregate_gmap_features, vilmodel.py:569-604)
        G = batch["gmap_step_ids"].shape[1]
        Tmax = max(step_lens)

        def gmap_weights():                                               # a host decision (hostsync): python loops + upload
            W = np.zeros((B, G, Tmax * Vmax), np.float32)      # (numpy: ~2000 element writes, each a torch op otherwise)
            vl = view_lens.cpu().tolist()
            for i in range(B):
                visited, unvisited = {}, {}
                for t in range(step_lens[i]):
                    n = vl[offs[i] + t]
                    visited[batch["traj_vpids"][i][t]] = (t, n)
                    for j, vp in enumerate(batch["traj_cand_vpids"][i][t]):
                        if vp not in visited:
                            unvisited.setdefault(vp, []).append((t, j))
                for k, vp in enumerate(batch["gmap_vpids"][i][1:]):
                    if vp in visited:
                        t, n = visited[vp]
                        W[i, k + 1, t * Vmax:t * Vmax + n] = 1.0 / n
                    else:
                        occ = unvisited[vp]
                        for t, j in occ:
                            W[i, k + 1, t * Vmax + j] += 1.0 / len(occ)
            return torch.from_numpy(W).to(dev)
        W = hs.host(gmap_weights)
        # (B, Tmax * Vmax, H) padded token blocks as ONE gather from [traj rows | a zero row] (host-built row index; a
        # per-episode pad + cat was 2 x 32 tiny launches per step with its backward)
        def pad_index():
            idx = np.full((B, Tmax * Vmax), traj.shape[0] * Vmax, dtype=np.int64)         # the zero row
            for i in range(B):
                n = step_lens[i] * Vmax
                idx[i, :n] = offs[i] * Vmax + np.arange(n)
            return torch.from_numpy(idx.reshape(-1)).to(dev)
        rows = torch.cat([traj.reshape(-1, H), traj.new_zeros(1, H)], 0)
        tok = rows.index_select(0, hs.host(pad_index)).view(B, Tmax * Vmax, H)
        gmap_img = torch.bmm(W, tok)                               # row 0 ([stop]) stays zero
        ge, le = b.global_encoder, b.local_encoder
        gmap_input = gmap_img + ge.gmap_step_embeddings(batch["gmap_step_ids"].long()) + ag.layer_norm(
            ag.linear(batch["gmap_pos_fts"].float(), ge.gmap_pos_embeddings[0].weight, ge.gmap_pos_embeddings[0].bias),
            ge.gmap_pos_embeddings[1])
        gmap_masks = _seq_masks(batch["gmap_lens"], G)

        # local branch input: last step's tokens behind a zero [stop] token (vp_input_embedding, vilmodel.py:545-560)
        last = hs.host(lambda: torch.tensor([offs[i + 1] - 1 for i in range(B)], device=dev))
        vp_lens = view_lens[last] + 1
        max_vp = hs.host(lambda: int(vp_lens.max()))
        vp_img = torch.cat([traj.new_zeros(B, 1, H), traj[last]], 1)[:, :max_vp]
        vp_input = vp_img + ag.layer_norm(
            ag.linear(batch["vp_pos_fts"].float(), le.vp_pos_embeddings[0].weight, le.vp_pos_embeddings[0].bias),
            le.vp_pos_embeddings[1])
        vp_masks = _seq_masks(vp_lens, max_vp)

        map_embeds = torch.cat([cells, gmap_input], 1)
        map_masks = torch.cat([cell_masks, gmap_masks], 1)
        map_embeds = VT.pre_ln_encoder(b, b.grid_encoder, map_embeds, map_masks)
        for layer in b.grid_txt_encoder.x_layers:
            xa = layer.visual_attention
            kv = VT._cat_linear(txt_embeds, [xa.att.key, xa.att.value], out_planes=0)
            map_embeds = VT.x_layer(b, layer, kv, txt_masks, map_embeds, map_masks)
        map_embeds, vp_input = hs.boundary(VT.CUT_MAP, map_embeds), hs.boundary(VT.CUT_MAP, vp_input)
        return dict(txt_embeds=txt_embeds, txt_masks=txt_masks, map_embeds=map_embeds, map_masks=map_masks,
                    gmap_masks=gmap_masks, vp_input=vp_input, vp_masks=vp_masks, last=last, view_lens=view_lens,
                    last_view_lens=only_view_lens[last], last_obj_lens=None if obj_lens is None else obj_lens[last])

    def _encode(self, batch):
        """GlocalTextPathCMT.forward (vilmodel.py:668-766) -> gmap_embeds, vp_embeds, gridmap_embeds, front."""
        b = self
        f = self._front(batch)
        H = f["map_embeds"].shape[-1]
        G = f["gmap_masks"].shape[1]
        gridmap_embeds = f["map_embeds"][:, N_CELLS:]
        kv_embeds = torch.cat([f["map_embeds"], f["txt_embeds"]], 1)
        kv_masks = torch.cat([f["map_masks"], f["txt_masks"]], 1)
        q = torch.cat([gridmap_embeds, f["vp_input"]], 1)
        q_masks = torch.cat([f["gmap_masks"], f["vp_masks"]], 1)
        xl = b.local_encoder.encoder.x_layers
        kv_all = VT._cat_linear(kv_embeds, [m for l in xl for m in (l.visual_attention.att.key,
                                                                    l.visual_attention.att.value)], out_planes=0)
        for layer, kv in zip(xl, ag.split_with_planes(kv_all, 2 * H)):       # (split: see vilmodel_train.encode_navigation)
            q = VT.x_layer(b, layer, kv, kv_masks, q, q_masks)
        return q[:, :G], q[:, G:], gridmap_embeds, f

    _ARGS = ("txt_ids", "txt_lens", "traj_view_img_fts", "traj_obj_img_fts", "traj_loc_fts", "traj_nav_types",
             "traj_step_lens", "traj_vp_view_lens", "traj_vp_obj_lens", "traj_vpids", "traj_cand_vpids", "gmap_lens",
             "gmap_step_ids", "gmap_pos_fts", "gmap_pair_dists", "gmap_vpids", "vp_pos_fts", "grid_fts", "grid_map")

    def forward(self, txt_ids, txt_lens, traj_view_img_fts, traj_obj_img_fts, traj_loc_fts, traj_nav_types,
                traj_step_lens, traj_vp_view_lens, traj_vp_obj_lens, traj_vpids, traj_cand_vpids, gmap_lens, gmap_step_ids,
                gmap_pos_fts, gmap_pair_dists, gmap_vpids, vp_pos_fts, grid_fts, grid_map, target_patch_id=None,
                gridmap_pos_fts=None, return_gmap_embeds=True):
        """The reference's positional surface (pretrain_src/model/vilmodel.py:668-766): returns (gmap_embeds,
        vp_embeds, map_embeds[:, n_cells:]) -- the last one is the [stop | nodes] part of the map sequence BEFORE the
        local encoder (what sap's grid head reads).  gmap_pair_dists / target_patch_id are accepted and unused, as in
        the reference."""
        vals = (txt_ids, txt_lens, traj_view_img_fts, traj_obj_img_fts, traj_loc_fts, traj_nav_types, traj_step_lens,
                traj_vp_view_lens, traj_vp_obj_lens, traj_vpids, traj_cand_vpids, gmap_lens, gmap_step_ids, gmap_pos_fts,
                gmap_pair_dists, gmap_vpids, vp_pos_fts, grid_fts, grid_map)
        batch = defaultdict(lambda: None, zip(self._ARGS, vals))
        batch["gridmap_pos_fts"] = gridmap_pos_fts
        gmap_embeds, vp_embeds, gridmap_embeds, _ = self._encode(batch)
        return (gmap_embeds if return_gmap_embeds else None), vp_embeds, gridmap_embeds

    def _mlm_text(self, f):
        """vilmodel.py:830-856: the text attends to [gmap | vp] through forward_lang2visn of every local layer."""
        vp_embeds = torch.cat([f["map_embeds"][:, N_CELLS:], f["vp_input"]], 1)
        vp_masks = torch.cat([f["gmap_masks"], f["vp_masks"]], 1)
        txt = f["txt_embeds"]
        for layer in self.local_encoder.encoder.x_layers:
            txt = VT.lang2visn_layer(self, layer, txt, f["txt_masks"], vp_embeds, vp_masks)
        return txt

    def forward_mlm(self, txt_ids, txt_lens, traj_view_img_fts, traj_obj_img_fts, traj_loc_fts, traj_nav_types,
                    traj_step_lens, traj_vp_view_lens, traj_vp_obj_lens, traj_vpids, traj_cand_vpids, gmap_lens,
                    gmap_step_ids, gmap_pos_fts, gmap_pair_dists, gmap_vpids, vp_pos_fts, grid_fts, grid_map,
                    gridmap_pos_fts):
        """pretrain_src/model/vilmodel.py:767-856: text embeddings (B, L, H) after the language-to-vision layers."""
        vals = (txt_ids, txt_lens, traj_view_img_fts, traj_obj_img_fts, traj_loc_fts, traj_nav_types, traj_step_lens,
                traj_vp_view_lens, traj_vp_obj_lens, traj_vpids, traj_cand_vpids, gmap_lens, gmap_step_ids, gmap_pos_fts,
                gmap_pair_dists, gmap_vpids, vp_pos_fts, grid_fts, grid_map)
        batch = defaultdict(lambda: None, zip(self._ARGS, vals))
        batch["gridmap_pos_fts"] = gridmap_pos_fts
        return self._mlm_text(self._front(batch))



class BertPredictionHeadTransform(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class BertLMPredictionHead(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.transform = BertPredictionHeadTransform(c)
        self.decoder = nn.Linear(c.hidden_size, c.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(c.vocab_size))


class BertOnlyMLMHead(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.predictions = BertLMPredictionHead(c)


class RegionClassification(nn.Module):
    def __init__(self, hidden_size, label_dim):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(hidden_size, hidden_size), nn.ReLU(),
                                 nn.LayerNorm(hidden_size, eps=1e-12), nn.Linear(hidden_size, label_dim))


def _seq_masks(lens, max_len=None):
    lens = lens.long()
    if max_len is None:
        max_len = hs.host(lambda: int(lens.max()))
    return torch.arange(max_len, device=lens.device).unsqueeze(0) < lens.unsqueeze(1)


def _gather_span(x, start, length, max_len):
    """pad_tensors_wgrad([x_b[start_b : start_b + length_b]]) -> (B, max_len, H), zero padded (data movement)."""
    B, S, H = x.shape
    p = torch.arange(max_len, device=x.device).unsqueeze(0).expand(B, max_len)
    idx = (start.unsqueeze(1) + p).clamp(max=S - 1)
    keep = p < length.unsqueeze(1)
    return x.gather(1, idx.unsqueeze(-1).expand(B, max_len, H)) * keep.unsqueeze(-1), keep


class GlocalTextPathCMTPreTraining(nn.Module):
    def __init__(self, config):
        super().__init__()
        c = self.config = config
        H = c.hidden_size
        self.bert = GlocalTextPathCMT(c)
        self.global_sap_head = V.ClsPrediction(H)
        tasks = c.pretrain_tasks
        if "mlm" in tasks:
            self.mlm_head = BertOnlyMLMHead(c)
        if "mrc" in tasks:
            self.image_classifier = RegionClassification(H, c.image_prob_size)
            self.obj_classifier = (RegionClassification(H, c.obj_prob_size)
                                   if c.obj_prob_size > 0 and c.obj_prob_size != c.image_prob_size else None)
        if "sap" in tasks:
            self.local_sap_head = V.ClsPrediction(H)
            self.grid_sap_head = V.ClsPrediction(H)
            self.sap_fuse_linear = V.ClsPrediction(H, input_size=H * 2) if c.glocal_fuse else None
        if "og" in tasks:
            self.og_head = V.ClsPrediction(H)
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)) and m.weight.dtype == torch.float32:
                nn.init.normal_(m.weight, std=0.02)
        self.tie_weights()

    def tie_weights(self):
        """pretrain_cmt.py:66-69: the MLM decoder shares the word-embedding matrix."""
        if "mlm" in self.config.pretrain_tasks:
            self.mlm_head.predictions.decoder.weight = self.bert.embeddings.word_embeddings.weight

    def load_state_dict(self, sd, strict=True):
        out = super().load_state_dict(sd, strict=strict)
        self.tie_weights()
        return out

    # the building blocks of vilmodel_train take `model` for .heads / .config / .training
    @property
    def heads(self):
        return self.bert.heads

    def _front(self, batch):
        return self.bert._front(batch)

    def _bert_forward(self, batch):
        return self.bert._encode(batch)

    # ---- tasks ---------------------------------------------------------------------------------------------------
    def forward(self, batch, task, compute_loss=True):
        batch = defaultdict(lambda: None, batch)
        if task.startswith("mlm"):
            return self.forward_mlm(batch, compute_loss)
        if task.startswith("mrc"):
            return self.forward_mrc(batch, compute_loss)
        if task.startswith("sap"):
            return self.forward_sap(batch, compute_loss)
        if task.startswith("og"):
            return self.forward_og(batch, compute_loss)
        raise ValueError("invalid task")

    def forward_mlm(self, batch, compute_loss=True):
        """pretrain_cmt.py:131-153 + vilmodel.py:767-856: the text attends to [gmap | vp] through
        forward_lang2visn of every local cross-modal layer, then the tied-decoder MLM head on masked tokens."""
        txt = self.bert._mlm_text(self._front(batch))
        labels = batch["txt_labels"]
        sel = labels != -1
        hidden = hs.select(txt, sel)                                          # only masked tokens
        p = self.mlm_head.predictions
        h = ag.layer_norm(ag.gelu(ag.linear(hidden, p.transform.dense.weight, p.transform.dense.bias)),
                          p.transform.LayerNorm)
        # decoder(h) + bias.  The vocabulary (30 522 rows) is not a multiple of 8, which the plane GEMMs and the row-major weight
        # gradient need: the tied matrix and its bias are extended by zero rows for the call (one cat each; the gradient of the
        # cat is the row slice) and the extra logit columns cut off -- instead of the fp32-A fallback kernels (5 calls of ~260 us
        # per mlm step).
        Nv = p.decoder.weight.shape[0]
        pad = (-Nv) % 8
        if pad and h.is_cuda and PAD_VOCAB:
            w = torch.cat([p.decoder.weight, p.decoder.weight.new_zeros(pad, p.decoder.weight.shape[1])], 0)
            b = torch.cat([p.bias, p.bias.new_zeros(pad)], 0)
            scores = ag.linear(h, w, b)[..., :Nv]
        else:
            scores = ag.linear(h, p.decoder.weight, p.bias)
        if compute_loss:
            return F.cross_entropy(scores, hs.select(labels, sel).long(), reduction="none")
        return scores

    @staticmethod
    def _region_head(head, x):
        net = head.net
        return ag.linear(ag.layer_norm(ag.relu(ag.linear(x, net[0].weight, net[0].bias)), net[2]), net[3].weight, net[3].bias)

    def forward_mrc(self, batch, compute_loss=True):
        """pretrain_cmt.py:161-213: soft-label classification of the masked views (and objects) of the last step."""
        _, vp_embeds, _, f = self._bert_forward(batch)
        B = vp_embeds.shape[0]
        one = torch.ones(B, dtype=torch.long, device=vp_embeds.device)
        masks = batch["vp_view_mrc_masks"].bool()
        view_embeds, _ = _gather_span(vp_embeds, one, f["last_view_lens"], masks.shape[1])      # [stop] at 0
        view_logits = self._region_head(self.image_classifier, hs.select(view_embeds, masks))
        view_targets = hs.select(batch["vp_view_probs"], masks)
        obj_logits = obj_targets = None
        if f["last_obj_lens"] is not None:
            omasks = batch["vp_obj_mrc_masks"].bool()
            obj_embeds, _ = _gather_span(vp_embeds, one + f["last_view_lens"], f["last_obj_lens"], omasks.shape[1])
            head = self.image_classifier if self.obj_classifier is None else self.obj_classifier
            obj_logits = self._region_head(head, hs.select(obj_embeds, omasks))
            obj_targets = hs.select(batch["vp_obj_probs"], omasks)
        if not compute_loss:
            return view_logits, view_targets, obj_logits, obj_targets
        loss = F.kl_div(F.log_softmax(view_logits, -1), view_targets, reduction="none").sum(1)
        if obj_logits is not None:
            loss = torch.cat([loss, F.kl_div(F.log_softmax(obj_logits, -1), obj_targets, reduction="none").sum(1)], 0)
        return loss

    def forward_og(self, batch, compute_loss=True):
        """pretrain_cmt.py:292-321: object grounding over the last step's object tokens."""
        _, vp_embeds, _, f = self._bert_forward(batch)
        B = vp_embeds.shape[0]
        one = torch.ones(B, dtype=torch.long, device=vp_embeds.device)
        max_obj = hs.host(lambda: int(f["last_obj_lens"].max()))
        obj_embeds, obj_masks = _gather_span(vp_embeds, one + f["last_view_lens"], f["last_obj_lens"], max_obj)
        obj_logits = VT.cls_head(self.og_head, obj_embeds).masked_fill(~obj_masks, -float("inf"))
        if compute_loss:
            return F.cross_entropy(obj_logits, batch["obj_labels"].long(), reduction="none")
        return obj_logits

    def forward_sap(self, batch, compute_loss=True):
        """pretrain_cmt.py:215-290."""
        gmap_embeds, vp_embeds, grid_embeds, f = self._bert_forward(batch)
        dev = gmap_embeds.device
        B, G = f["gmap_masks"].shape
        Vp = vp_embeds.shape[1]
        fuse_raw = None
        if self.sap_fuse_linear is not None:
            fuse_raw = VT.cls_head(self.sap_fuse_linear, torch.cat([gmap_embeds[:, 0], vp_embeds[:, 0]], 1))
        g_raw = VT.cls_head(self.global_sap_head, gmap_embeds)
        grid_raw = VT.cls_head(self.grid_sap_head, grid_embeds)
        l_raw = VT.cls_head(self.local_sap_head, vp_embeds)
        # navigable = nav_type 1 in the LAST step (:237-243); [stop] always allowed
        nav_types = batch["traj_nav_types"][f["last"]]
        nav = torch.cat([torch.ones(B, 1, dtype=torch.bool, device=dev), nav_types == 1], 1)[:, :Vp]
        visited = batch["gmap_visited_masks"].bool()
        def index_maps():                                                 # a host decision (hostsync): vpid loops + upload
            cand_of_node = np.full((B, G), -2, np.int32)
            cand_visited = np.zeros((B, Vp), np.uint8)
            vis_host = visited.cpu().numpy()
            for i in range(B):
                vset = set(vp for vp, m in zip(batch["gmap_vpids"][i], vis_host[i]) if m)
                tmp = {}
                for j, cv in enumerate(batch["traj_cand_vpids"][i][-1]):
                    if cv in vset:
                        cand_visited[i, j + 1] = 1
                    else:
                        tmp[cv] = j + 1
                for j, vp in enumerate(batch["gmap_vpids"][i]):
                    if j > 0 and vp not in vset:
                        cand_of_node[i, j] = tmp.get(vp, -1)
            return torch.from_numpy(cand_of_node).to(dev), torch.from_numpy(cand_visited).to(dev)
        cand_of_node, cand_visited = hs.host(index_maps)
        global_logits, local_logits, grid_logits, fused_logits = VT.fuse_logits(
            g_raw, l_raw, grid_raw, fuse_raw, f["gmap_masks"], visited, nav, cand_of_node, cand_visited)
        if not compute_loss:
            return global_logits, local_logits, fused_logits, batch["global_act_labels"], batch["local_act_labels"]
        gl, ll = batch["global_act_labels"].long(), batch["local_act_labels"].long()
        losses = [F.cross_entropy(global_logits, gl, reduction="none"), F.cross_entropy(local_logits, ll, reduction="none"),
                  F.cross_entropy(fused_logits, gl, reduction="none"), F.cross_entropy(grid_logits, gl, reduction="none")]
        n_stop, n_go = hs.host(lambda: (int((gl == 0).sum()), int((gl != 0).sum())))
        stop_rate = n_stop / n_go if n_go != 0 else 1.0                       # :278-281: stop samples are re-weighted
        out = 0
        for loss, lab in zip(losses, (gl, ll, gl, gl)):
            out = out + (torch.where(lab == 0, loss / stop_rate, loss) if stop_rate != 0 else loss)
        return out
