"""CLIP ViT-B/32 patch encoder on the hand-written HIP kernels -- the producer of the VLN-CE grid memory.

Reference: /root/reference/VLN_CE/vlnce_baselines/models/gridmap/clip.py (VisionTransformer :69-101, ResidualAttentionBlock
:31-57, QuickGELU :26-28), instantiated as GlocalTextPathNavCMT.clip = CLIP(224, 32, 768, 12, 12) (gridmap/vilmodel.py:
627-629) and called once per step on the 12 view images of every episode (Policy_ViewSelection_GridMap.py:323-344); its
(B*12, 50, 768) tokens then travel GPU -> numpy -> per-episode python lists -> torch.tensor(...).cuda() (:340-357, 496).
Here the module tree only HOLDS the parameters under the reference's state_dict keys (`visual.conv1.weight`,
`visual.transformer.resblocks.N.attn.in_proj_weight`, ...); the arithmetic is
    patchify (im2col + MFMA GEMM, K = 3*32*32) -> +class/positional embedding -> ln_pre
    12 x [ ln_1 -> QKV GEMM -> attention_rows (no mask) -> out_proj (+residual) -> ln_2 -> c_fc + QuickGELU -> c_proj (+res) ]
    -> ln_post
on gridmm_linear_planes / gridmm_attention_rows / gridmm_layernorm, and `encode_into()` writes the 49 patch tokens of
every view as fp16 straight into the grid memory's next slot (gridmm_tokens_to_slab): no host round trip.
Inference only (the reference runs the tower under torch.no_grad(), Policy_ViewSelection_GridMap.py:335).
"""
import torch
from torch import nn

from . import ops
from .vilmodel import MultiheadAttentionParams


class _Mlp(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.c_fc = nn.Linear(d, 4 * d)
        self.c_proj = nn.Linear(4 * d, d)          # ("gelu" = QuickGELU has no parameters)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head):
        super().__init__()
        self.attn = MultiheadAttentionParams(d_model)
        self.ln_1 = nn.LayerNorm(d_model)
        self.mlp = _Mlp(d_model)
        self.ln_2 = nn.LayerNorm(d_model)
        self.n_head = n_head


class Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads) for _ in range(layers)])


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads):
        super().__init__()
        assert width == heads * 64, "the attention kernels are built for head_dim 64 (CLIP-B: 768 = 12 x 64)"
        self.input_resolution, self.patch_size, self.layers, self.heads = input_resolution, patch_size, layers, heads
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self._packed = {}

    def _pack(self, key, w, b):
        ver = (w.data_ptr(), w._version, None if b is None else (b.data_ptr(), b._version))
        ent = self._packed.get(key)
        if ent is None or ent[0] != ver:
            ent = (ver, ops.PackedLinear(w.reshape(w.shape[0], -1), b))
            self._packed[key] = ent
        return ent[1]

    @torch.no_grad()
    def forward(self, x):
        """x: (N, 3, R, R) normalised images on the GPU -> (N, 1 + (R/P)^2, width) fp32 tokens (ln_post applied)."""
        N, P = x.shape[0], self.patch_size
        g = self.input_resolution // P
        W = self.conv1.weight.shape[0]
        # im2col: (N, 3, g, P, g, P) -> (N * g * g, 3 * P * P), column order (c, kh, kw) = conv1.weight.view(W, -1)
        cols = x.float().reshape(N, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(N * g * g, 3 * P * P)
        patches = ops.linear(cols, self._pack("conv1", self.conv1.weight, None)).f32.view(N, g * g, W)
        tok = torch.empty(N, g * g + 1, W, dtype=torch.float32, device=x.device)
        tok[:, 0] = self.class_embedding
        tok[:, 1:] = patches
        tok += self.positional_embedding
        xs = ops.layernorm(tok, self.ln_pre.weight, self.ln_pre.bias, self.ln_pre.eps).f32
        for i, blk in enumerate(self.transformer.resblocks):
            k = "blk%d" % i
            h = ops.layernorm(xs, blk.ln_1.weight, blk.ln_1.bias, blk.ln_1.eps, want_f32=False, want_planes=True)
            qkv = ops.linear(h, self._pack(k + ".in", blk.attn.in_proj_weight, blk.attn.in_proj_bias), want_f32=False,
                             want_planes=True)
            sl = lambda c0: (qkv.hi[..., c0:c0 + W], qkv.lo[..., c0:c0 + W])
            ctx = ops.attention_rows(sl(0), sl(W), sl(2 * W), None, heads=self.heads)
            xs = ops.linear(ctx, self._pack(k + ".o", blk.attn.out_proj.weight, blk.attn.out_proj.bias), residual=xs).f32
            h = ops.layernorm(xs, blk.ln_2.weight, blk.ln_2.bias, blk.ln_2.eps, want_f32=False, want_planes=True)
            f = ops.linear(h, self._pack(k + ".fc", blk.mlp.c_fc.weight, blk.mlp.c_fc.bias), act=ops.ACT_QUICKGELU,
                           want_f32=False, want_planes=True)
            xs = ops.linear(f, self._pack(k + ".pr", blk.mlp.c_proj.weight, blk.mlp.c_proj.bias), residual=xs).f32
        return ops.layernorm(xs, self.ln_post.weight, self.ln_post.bias, self.ln_post.eps).f32


class CLIP(nn.Module):
    """gridmap/clip.py:103-113: `visual` only."""

    def __init__(self, input_resolution=224, patch_size=32, width=768, layers=12, heads=12):
        super().__init__()
        self.visual = VisionTransformer(input_resolution, patch_size, width, layers, heads)

    def forward(self, x):
        return self.visual(x)

    @torch.no_grad()
    def encode_into(self, images, slot, n_views=12):
        """images (B * n_views, 3, R, R) -> the patch tokens of every view, fp16, into `slot` = the (B, n_views * 49, width)
        view GridMemoryBatch.next_slot() hands out; returns the fp32 tokens incl. the class token, (B * n_views, 50, width)."""
        tok = self.visual(images)
        ops.tokens_to_slab(tok, slot, n_views)
        return tok
