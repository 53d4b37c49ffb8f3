"""VLN-CE twin of the navigation model: same backbone and kernels, habitat-side calling convention.

Reference: /root/reference/VLN_CE/vlnce_baselines/models/gridmap/vilmodel.py
  GlocalTextPathNavCMT.forward(mode, batch)             :802-818   batches are positional TUPLES
  forward_text :678-682   forward_panorama_per_step :684-708 (no object tokens)
  forward_navigation_per_step                            :710-800   grid aggregation + encoders as in the discrete model,
      then  fused = global_sap_head(gmap)*w + local_sap_head(vp)*(1-w),  both truncated to max(candidate_lengths) and
      masked by vp_nav_masks -- no visited masks, no vpid-keyed fusion, grid_sap_head unused; returns ONLY fused_logits.
The reference module also owns a CLIP-B/32 and a ViT-B/16 tower (`clip.*`, `visual_encoder.*`, :627-631) that feed the
grid / the views on the fly.  `clip` -- the producer of the grid memory -- is built here on the HIP
kernels when the config asks for it (`with_clip_tower=True`; clip_encoder.CLIP, same state_dict keys) and writes its
patch tokens straight into the memory's next slot (`encode_observation`); `visual_encoder` (the timm ViT-B/16 that makes
the 768-d VIEW features) is outside §8 and its checkpoint keys are ignored on load.
The grid memory for this variant is GridMemoryBatch(geom=synthetic.VLNCE_R2R / VLNCE_RXR).
"""
import torch

from . import ops
from .vilmodel import GlocalTextPathNavCMT as _DiscreteNavCMT, default_config  # noqa: F401


class GlocalTextPathNavCMT(_DiscreteNavCMT):
    def __init__(self, config=None):
        super().__init__(config)
        self.global_encoder.sprel_linear = None            # gridmap/vilmodel.py:575
        if getattr(self.config, "with_clip_tower", False):
            from .clip_encoder import CLIP
            self.clip = CLIP(input_resolution=224, patch_size=32, width=768, layers=12, heads=12)   # :627-629

    def load_state_dict(self, sd, strict=True):
        drop = ("visual_encoder.",) if hasattr(self, "clip") else ("clip.", "visual_encoder.")
        sd = {k: v for k, v in sd.items() if not k.startswith(drop)}
        return super().load_state_dict(sd, strict=strict)

    @torch.no_grad()
    def encode_observation(self, grid_images, grid_memory):
        """Policy_ViewSelection_GridMap.py:335-357 on the device: the (B * 12, 3, 224, 224) normalised view images go
        through the CLIP tower and their 49 patch tokens per view are written, fp16, into the memory's next slot --
        what `grid_memory.step(depth, feats=None, ...)` then projects and bins.  Returns the (B * 12, 50, 768) tokens."""
        return self.clip.encode_into(grid_images, grid_memory.next_slot(), n_views=grid_memory.geom.n_views)

    def forward_navigation_per_step(self, txt_embeds, txt_masks, gmap_img_embeds, gmap_step_ids, gmap_pos_fts,
                                    gmap_masks, vp_img_embeds, vp_pos_fts, vp_masks, vp_nav_masks, grid_fts,
                                    grid_map_indexs, gridmap_pos_fts, candidate_lengths, grid_memory=None):
        """gridmap/vilmodel.py:710-800 -> fused_logits (B, max(candidate_lengths))."""
        C = int(max(candidate_lengths))
        nav = vp_nav_masks[:, :C]
        if self._differentiable():
            from . import vilmodel_train as VT
            cells, cell_masks = VT.grid_cells(self, txt_embeds.float(), grid_fts, grid_map_indexs, gridmap_pos_fts,
                                              grid_memory)
            gmap_out, vp_out, _ = VT.encode_navigation(self, txt_embeds.float(), txt_masks.bool(), cells, cell_masks,
                                                       gmap_img_embeds, gmap_step_ids, gmap_pos_fts, gmap_masks.bool(),
                                                       vp_img_embeds, vp_pos_fts, vp_masks.bool())
            fw = torch.sigmoid(VT.cls_head(self.sap_fuse_linear, torch.cat([gmap_out[:, 0], vp_out[:, 0]], 1))).unsqueeze(1)
            g = (VT.cls_head(self.global_sap_head, gmap_out) * fw)[:, :C]
            l = (VT.cls_head(self.local_sap_head, vp_out) * (1 - fw))[:, :C]
            ninf = -float("inf")
            return g.masked_fill(~nav.bool(), ninf) + l.masked_fill(~nav.bool(), ninf)
        with torch.no_grad():
            dev = txt_embeds.device
            B = txt_embeds.shape[0]
            gmap_out, vp_out, _ = self._encode_navigation_infer(
                txt_embeds, txt_masks, gmap_img_embeds, gmap_step_ids, gmap_pos_fts, gmap_masks, vp_img_embeds,
                vp_pos_fts, vp_masks, grid_fts, grid_map_indexs, gridmap_pos_fts, grid_memory)
            fuse_raw = self._cls(self.sap_fuse_linear, "fuse", torch.cat([gmap_out[:, 0], vp_out[:, 0]], 1))
            g_raw = self._cls(self.global_sap_head, "ghead", gmap_out)[:, :C].contiguous()
            l_raw = self._cls(self.local_sap_head, "lhead", vp_out)[:, :C].contiguous()
            # fused[j] = global[j] + local[j]: the fusion kernel with the identity candidate map and nothing visited
            nav_u8 = self._u8(nav)
            ident = torch.arange(C, dtype=torch.int32, device=dev).unsqueeze(0).expand(B, C).contiguous()
            zeros = torch.zeros(B, C, dtype=torch.uint8, device=dev)
            _, _, _, fused = ops.fuse_logits(g_raw, l_raw, g_raw, fuse_raw, nav_u8, zeros, nav_u8, ident, zeros)
            return fused

    def forward(self, mode, batch, **kwargs):
        """gridmap/vilmodel.py:802-818 (tuple batches)."""
        if mode == "language":
            return self.forward_text(batch[0], batch[1])
        if mode == "panorama":
            view_img_fts, loc_fts, nav_types, view_lens = batch
            return self.forward_panorama_per_step(view_img_fts, None, loc_fts, nav_types, view_lens, None)
        if mode == "navigation":
            return self.forward_navigation_per_step(*batch, **kwargs)
        raise NotImplementedError("wrong mode: %s" % mode)
