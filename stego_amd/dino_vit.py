"""Frozen DINO ViT backbone for DinoFeaturizer (stock PyTorch-ROCm; NOT part of the native hot path).

Role of the reference's ``src/dino/vision_transformer.py:135-277``: a timm-style ViT
(tiny/small/base, patch 8 or 16) whose ``get_intermediate_feat(x, n)`` hands back the
normalised token features (and the raw qkv) of the last ``n`` blocks.  Parameter names match
the DINO checkpoints (``cls_token, pos_embed, patch_embed.proj.*, blocks.i.{norm1,attn.qkv,
attn.proj,norm2,mlp.fc1,mlp.fc2}.*, norm.*``) so ``load_state_dict`` of a released checkpoint
works unchanged.  Differences in HOW: attention goes through
``F.scaled_dot_product_attention`` and the [B,heads,N,N] attention maps the reference
materialises at every block (:232-236) are only built when a caller asks for them.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, need_attn=False):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = None
        if need_attn:
            attn = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)
            y = attn @ v
        else:
            y = F.scaled_dot_product_attention(q, k, v)
        return self.proj(y.transpose(1, 2).reshape(B, N, C)), attn, qkv


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads, qkv_bias)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x, need_attn=False):
        y, attn, qkv = self.attn(self.norm1(x), need_attn)
        x = x + y
        x = x + self.mlp(self.norm2(x))
        return x, attn, qkv


class VisionTransformer(nn.Module):
    def __init__(self, img_size=(224,), patch_size=16, in_chans=3, num_classes=0, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, eps=1e-6, **_):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size[0], patch_size, in_chans, embed_dim)
        n = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias, eps) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=eps)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init)

    @staticmethod
    def _init(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)

    def interpolate_pos_encoding(self, x, w, h):
        """Bicubic resize of the patch position table for inputs other than the training size
        (reference :176-196, incl. its +0.1 guard against the interpolate rounding issue)."""
        npatch, N = x.shape[1] - 1, self.pos_embed.shape[1] - 1
        if npatch == N and w == h:
            return self.pos_embed
        dim = x.shape[-1]
        side = int(math.sqrt(N))
        w0, h0 = w // self.patch_embed.patch_size + 0.1, h // self.patch_embed.patch_size + 0.1
        grid = self.pos_embed[:, 1:].reshape(1, side, side, dim).permute(0, 3, 1, 2)
        grid = F.interpolate(grid, scale_factor=(w0 / side, h0 / side), mode="bicubic")
        assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
        return torch.cat((self.pos_embed[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)

    def prepare_tokens(self, x):
        B, _, w, h = x.shape
        tok = self.patch_embed(x)
        tok = torch.cat((self.cls_token.expand(B, -1, -1), tok), dim=1)
        return tok + self.interpolate_pos_encoding(tok, w, h)

    def forward(self, x):
        return self.forward_feats(x)[:, 0]

    def forward_feats(self, x):
        x = self.prepare_tokens(x)
        for blk in self.blocks:
            x = blk(x)[0]
        return self.norm(x)

    def get_intermediate_feat(self, x, n=1, need_attn=False):
        """(feats, attns, qkvs) of the last n blocks; attns entries are None unless need_attn."""
        x = self.prepare_tokens(x)
        feats, attns, qkvs = [], [], []
        depth = len(self.blocks)
        for i, blk in enumerate(self.blocks):
            last = depth - i <= n
            x, attn, qkv = blk(x, need_attn and last)
            if last:
                feats.append(self.norm(x))
                attns.append(attn)
                qkvs.append(qkv)
        return feats, attns, qkvs

    def get_intermediate_layers(self, x, n=1):
        return self.get_intermediate_feat(x, n)[0]


def vit_tiny(patch_size=16, **kw):
    return VisionTransformer(patch_size=patch_size, embed_dim=192, depth=12, num_heads=3, **kw)


def vit_small(patch_size=16, **kw):
    return VisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6, **kw)


def vit_base(patch_size=16, **kw):
    return VisionTransformer(patch_size=patch_size, embed_dim=768, depth=12, num_heads=12, **kw)


ARCHS = {"vit_tiny": vit_tiny, "vit_small": vit_small, "vit_base": vit_base}
