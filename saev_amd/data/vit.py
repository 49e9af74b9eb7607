"""A plain pre-norm vision transformer in stock PyTorch (patch embedding, CLS token, learned positions, MHA + MLP blocks).

Only here so that the extraction hand-off (extract.py) has something to hook offline: there is no network for
checkpoints, so BASELINE.json configs[4] runs a randomly initialised model of DINOv2 ViT-L/14's shape
(``VisionTransformer.vit_l14()``: 24 blocks, d_model 1024, 16 heads, 14 x 14 patches of a 224 x 224 image -> 256 content
tokens + CLS).  The forward is PyTorch-ROCm's (``F.scaled_dot_product_attention``, ``nn.Linear``); nothing in it is this
package's hot path.  With real weights a user passes their own module and its list of blocks to ActivationRecorder.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F


class Block(torch.nn.Module):
    def __init__(self, d: int, heads: int, mlp_ratio: float = 4.0):
        super().__init__()
        self.heads = heads
        self.norm1, self.norm2 = torch.nn.LayerNorm(d, eps=1e-6), torch.nn.LayerNorm(d, eps=1e-6)
        self.qkv, self.proj = torch.nn.Linear(d, 3 * d), torch.nn.Linear(d, d)
        self.fc1, self.fc2 = torch.nn.Linear(d, int(mlp_ratio * d)), torch.nn.Linear(int(mlp_ratio * d), d)

    def forward(self, x):
        b, t, d = x.shape
        q, k, v = self.qkv(self.norm1(x)).view(b, t, 3, self.heads, d // self.heads).permute(2, 0, 3, 1, 4)
        x = x + self.proj(F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, t, d))
        return x + self.fc2(F.gelu(self.fc1(self.norm2(x))))


class VisionTransformer(torch.nn.Module):
    def __init__(self, *, d_model: int, depth: int, heads: int, patch: int, image: int, channels: int = 3):
        super().__init__()
        self.patch, self.n_patches = patch, (image // patch) ** 2
        self.embed = torch.nn.Conv2d(channels, d_model, kernel_size=patch, stride=patch)
        self.cls = torch.nn.Parameter(torch.zeros(1, 1, d_model))
        self.pos = torch.nn.Parameter(0.02 * torch.randn(1, self.n_patches + 1, d_model))
        self.blocks = torch.nn.ModuleList(Block(d_model, heads) for _ in range(depth))
        self.norm = torch.nn.LayerNorm(d_model, eps=1e-6)

    @classmethod
    def vit_l14(cls) -> "VisionTransformer":
        return cls(d_model=1024, depth=24, heads=16, patch=14, image=224)

    def forward(self, images):
        x = self.embed(images).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls.expand(x.shape[0], -1, -1), x], dim=1) + self.pos
        for blk in self.blocks:
            x = blk(x)
        return self.norm(x)
