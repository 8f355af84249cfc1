"""CPU restatement of the reference ControlNet (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows controlnet/controlnet.py: conditioning embedder :62-104, constructor :179-438 (SD-1.5 defaults
:181-217), forward :662-881 (skip flags :802-811, cond add :816-817, down loop :820-833, mid :836-846,
zero convs :850-858, scaling :861-874); MultiControlNetModel: controlnet/multicontrolnet.py:45-99.
Module names reproduce the reference's state-dict keys.  Only the configuration the hot path reaches is
restated (no class/addition embeddings, rgb channel order, no attention mask).
"""
import torch
import torch.nn.functional as F
from torch import nn

from .blocks import Timesteps, TimestepEmbedding, get_down_block, UNetMidBlock2DCrossAttn


class CondEmbedding(nn.Module):
    """ControlNetConditioningEmbedding (controlnet/controlnet.py:62-104)"""

    def __init__(self, out_channels, cond_channels=3, widths=(16, 32, 96, 256)):
        super().__init__()
        self.conv_in = nn.Conv2d(cond_channels, widths[0], 3, padding=1)
        self.blocks = nn.ModuleList()
        for a, b in zip(widths[:-1], widths[1:]):
            self.blocks.append(nn.Conv2d(a, a, 3, padding=1))
            self.blocks.append(nn.Conv2d(a, b, 3, padding=1, stride=2))
        self.conv_out = nn.Conv2d(widths[-1], out_channels, 3, padding=1)   # zero-initialised in the reference

    def forward(self, c):
        e = F.silu(self.conv_in(c))
        for blk in self.blocks:
            e = F.silu(blk(e))
        return self.conv_out(e)


class ControlNetOracle(nn.Module):
    def __init__(self, in_channels=4, conditioning_channels=3, block_out_channels=(320, 640, 1280, 1280),
                 down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",), layers_per_block=2,
                 num_attention_heads=8, cross_attention_dim=768, norm_eps=1e-5, norm_num_groups=32,
                 conditioning_embedding_out_channels=(16, 32, 96, 256), global_pool_conditions=False):
        super().__init__()
        self.global_pool_conditions = global_pool_conditions
        c0 = block_out_channels[0]
        temb_dim = c0 * 4
        self.conv_in = nn.Conv2d(in_channels, c0, 3, padding=1)
        self.time_proj = Timesteps(c0, True, 0)
        self.time_embedding = TimestepEmbedding(c0, temb_dim)
        self.controlnet_cond_embedding = CondEmbedding(c0, conditioning_channels, conditioning_embedding_out_channels)
        self.down_blocks = nn.ModuleList()
        self.controlnet_down_blocks = nn.ModuleList([nn.Conv2d(c0, c0, 1)])
        out_c = c0
        for i, kind in enumerate(down_block_types):
            in_c, out_c = out_c, block_out_channels[i]
            last = i == len(block_out_channels) - 1
            self.down_blocks.append(get_down_block(
                kind, num_layers=layers_per_block, in_channels=in_c, out_channels=out_c, temb_channels=temb_dim,
                add_downsample=not last, resnet_eps=norm_eps, resnet_groups=norm_num_groups,
                cross_attention_dim=cross_attention_dim, num_attention_heads=num_attention_heads))
            for _ in range(layers_per_block + (0 if last else 1)):
                self.controlnet_down_blocks.append(nn.Conv2d(out_c, out_c, 1))
        self.controlnet_mid_block = nn.Conv2d(out_c, out_c, 1)
        self.mid_block = UNetMidBlock2DCrossAttn(in_channels=out_c, temb_channels=temb_dim, resnet_eps=norm_eps,
                                                 cross_attention_dim=cross_attention_dim,
                                                 num_attention_heads=num_attention_heads, resnet_groups=norm_num_groups)

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0,
                guess_mode=False, return_dict=False, skip_conv_in=False, skip_time_emb=False, **ignored):
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float64 if isinstance(t, float) else torch.int64)
        elif t.dim() == 0:
            t = t[None]
        t = t.expand(sample.shape[0])
        emb = self.time_embedding(self.time_proj(t).to(sample.dtype))
        x = self.conv_in(sample)
        if skip_conv_in:
            x = torch.zeros_like(x)
        if skip_time_emb:
            emb = torch.zeros_like(emb)
        x = x + self.controlnet_cond_embedding(controlnet_cond)
        res = (x,)
        for blk in self.down_blocks:
            if getattr(blk, "has_cross_attention", False):
                x, r = blk(hidden_states=x, temb=emb, encoder_hidden_states=encoder_hidden_states)
            else:
                x, r = blk(hidden_states=x, temb=emb)
            res += r
        x = self.mid_block(x, emb, encoder_hidden_states=encoder_hidden_states)
        down = [conv(r) for r, conv in zip(res, self.controlnet_down_blocks)]
        mid = self.controlnet_mid_block(x)
        if guess_mode and not self.global_pool_conditions:
            scales = torch.logspace(-1, 0, len(down) + 1) * conditioning_scale
            down = [d * s for d, s in zip(down, scales)]
            mid = mid * scales[-1]
        else:
            down = [d * conditioning_scale for d in down]
            mid = mid * conditioning_scale
        if self.global_pool_conditions:
            down = [d.mean(dim=(2, 3), keepdim=True) for d in down]
            mid = mid.mean(dim=(2, 3), keepdim=True)
        return down, mid


class MultiControlNetOracle(nn.Module):
    """controlnet/multicontrolnet.py:45-99 -- returns per-net lists (the sum is the router's job)."""

    def __init__(self, nets):
        super().__init__()
        self.nets = nn.ModuleList(nets)

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale, **kw):
        downs, mids = [], []
        for image, scale, net in zip(controlnet_cond, conditioning_scale, self.nets):
            d, m = net(sample, timestep, encoder_hidden_states, image, scale, **kw)
            downs.append(d)
            mids.append(m)
        return downs, mids
