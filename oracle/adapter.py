"""CPU restatement of the Ctrl-Adapter modules (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows model/resnet_block_2d.py:61-221 (ResnetBlock2D with output_size forwarded to the up-sampler),
model/adapter_spatial_temporal.py:11-292 (AdapterSpatioTemporal) and model/ctrl_adapter.py:17-224
(ControlNetAdapter).  Module names reproduce the reference's state-dict keys.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .blocks import (Timesteps, TimestepEmbedding, TemporalResnetBlock, BasicTransformerBlock,
                     TemporalBasicTransformerBlock, AlphaBlender, Upsample2D)


class AdapterResnet2D(nn.Module):
    """model/resnet_block_2d.py:61-221 as the adapter builds it (:83-92): eps 1e-6, swish, default temb norm,
    use_in_shortcut=True (1x1 shortcut always), optional nearest up-sampling of BOTH branches (:174-184)."""

    def __init__(self, in_channels, out_channels, temb_channels, eps=1e-6, up=False):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(32, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.upsample = Upsample2D(in_channels, use_conv=False) if up else None
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1)

    def forward(self, x, temb, output_size=None):
        h = F.silu(self.norm1(x))
        if self.upsample is not None:
            x = self.upsample(x, output_size)
            h = self.upsample(h, output_size)
        h = self.conv1(h) + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(self.dropout(F.silu(self.norm2(h))))
        return self.conv_shortcut(x) + h


class AdapterBlockOracle(nn.Module):
    """AdapterSpatioTemporal (model/adapter_spatial_temporal.py:11-292)"""

    def __init__(self, in_channels, out_channels, num_layers=1, add_spatial_resnet=True, add_temporal_resnet=True,
                 add_spatial_transformer=True, add_temporal_transformer=True, eps=1e-6, merge_factor=0.5,
                 up_sampling_scale=1.0, cross_attention_dim=1024, num_attention_heads=8, attention_head_dim=64):
        super().__init__()
        self.heads = in_channels // attention_head_dim           # :42 (head count used by the blocks)
        self.num_layers = num_layers
        self.up = up_sampling_scale
        self.sr, self.tr, self.st, self.tt = add_spatial_resnet, add_temporal_resnet, add_spatial_transformer, add_temporal_transformer
        if self.sr or self.tr:
            self.resnet_time_proj = Timesteps(out_channels, True, 0)
            self.resnet_time_embedding = TimestepEmbedding(in_channels, in_channels)
        if self.st or self.tt:
            self.norm = nn.GroupNorm(32, in_channels, eps=1e-6)
            self.inner_dim = num_attention_heads * attention_head_dim      # 512 regardless of C (:62)
            if self.tt:
                self.transformer_time_embedding = TimestepEmbedding(in_channels, self.inner_dim)
                self.transformer_time_proj = Timesteps(in_channels, True, 0)
            self.proj_in = nn.Linear(in_channels, self.inner_dim)
            self.proj_out = nn.Linear(self.inner_dim, in_channels)
        if self.sr:
            self.spatial_resnets = nn.ModuleList([
                AdapterResnet2D(in_channels, out_channels, in_channels, eps, up=(i == 0 and self.up > 1))
                for i in range(num_layers)])
        if self.tr:
            self.temporal_resnets = nn.ModuleList([
                TemporalResnetBlock(out_channels if self.sr else in_channels, out_channels, in_channels, eps)
                for _ in range(num_layers)])
        if self.st:
            self.spatial_attentions = nn.ModuleList([
                BasicTransformerBlock(self.inner_dim, self.heads, attention_head_dim, cross_attention_dim)
                for _ in range(num_layers)])
        if self.tt:
            self.temporal_attentions = nn.ModuleList([
                TemporalBasicTransformerBlock(self.inner_dim, self.inner_dim, self.heads, attention_head_dim,
                                              cross_attention_dim) for _ in range(num_layers)])
        if self.sr and self.tr:
            self.resnets_time_mixer = nn.ModuleList([AlphaBlender(merge_factor) for _ in range(num_layers)])
        if self.st and self.tt:
            self.transformers_time_mixer = nn.ModuleList([AlphaBlender(merge_factor) for _ in range(num_layers)])

    def forward(self, hidden_states, num_frames, timestep=None, encoder_hidden_states=None, sparsity_masking=None):
        bf, c, h, w = hidden_states.shape
        b = bf // num_frames
        # timestep normalisation (:190-198)
        if isinstance(timestep, (int, float)):
            timestep = torch.tensor([float(timestep)]).repeat_interleave(bf, dim=0)
        elif timestep.dim() == 0:
            timestep = timestep.reshape(1).float().repeat_interleave(bf, dim=0)
        elif timestep.dim() == 1 and len(timestep) == 1:
            timestep = timestep.float().repeat_interleave(bf, dim=0)
        elif timestep.dim() == 2:
            timestep = timestep.squeeze()
        timestep = timestep.to(hidden_states.dtype)
        indicator = torch.zeros(b, num_frames, dtype=hidden_states.dtype)
        x = hidden_states
        for i in range(self.num_layers):
            if self.sr or self.tr:
                temb = self.resnet_time_embedding(self.resnet_time_proj(timestep)).to(x.dtype)
            if self.sr:
                hh, ww = x.shape[2:]
                size = (int(hh * self.up), int(ww * self.up)) if i == 0 else None
                x = self.spatial_resnets[i](x, temb, output_size=size)
                h, w = x.shape[2:]
                if self.tr:
                    x_mix = x[None].reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
            if self.tr:
                x = x[None].reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
                x = self.temporal_resnets[i](x, temb.reshape(b, num_frames, -1))
                if self.sr:
                    x = self.resnets_time_mixer[i](x_spatial=x_mix, x_temporal=x, image_only_indicator=indicator)
                x = x.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)
            if not self.sr and not self.tr and i == 0 and self.up > 1:
                x = F.interpolate(x, scale_factor=self.up, mode="nearest")
                h, w = x.shape[2:]
            if self.st or self.tt:
                ehs = encoder_hidden_states
                if ehs.dim() == 2:
                    ehs = ehs.unsqueeze(1)
                if ehs.shape[0] == 1:
                    ehs = ehs.repeat_interleave(bf, dim=0)
                if self.tt:
                    first = ehs[None].reshape(b, num_frames, -1, ehs.shape[-1])[:, 0]
                    tctx = first[None].broadcast_to(h * w, b, 1, ehs.shape[-1]).reshape(h * w * b, 1, ehs.shape[-1])
                residual = x
                tok = self.norm(x).permute(0, 2, 3, 1).reshape(bf, h * w, c)
                tok = self.proj_in(tok)
                if self.tt:
                    fidx = torch.arange(num_frames).repeat(b, 1).reshape(-1)
                    femb = self.transformer_time_embedding(self.transformer_time_proj(fidx).to(tok.dtype))[:, None, :]
            if self.st:
                tok = self.spatial_attentions[i](tok, encoder_hidden_states=ehs)
                tok_mix = tok
            if self.tt:
                tok = tok + femb
                tok = self.temporal_attentions[i](tok, num_frames=num_frames, encoder_hidden_states=tctx)
                if self.st:
                    tok = self.transformers_time_mixer[i](x_spatial=tok_mix, x_temporal=tok, image_only_indicator=indicator)
            if self.st or self.tt:
                tok = self.proj_out(tok)
                x = tok.reshape(bf, h, w, c).permute(0, 3, 1, 2).contiguous() + residual
        return x


LOCATION_SLOTS = {"A": {3: [0, 1, 2], 2: [0, 2], 1: [2]}, "B": {3: [3, 4, 5], 2: [3, 5], 1: [5]},
                  "C": {3: [6, 7, 8], 2: [6, 8], 1: [8]}, "D": {3: [9, 10, 11], 2: [9, 11], 1: [11]}}
LOCATION_CHANNELS = {"A": {3: [320] * 3, 2: [320] * 2, 1: [320]}, "B": {3: [320, 640, 640], 2: [320, 640], 1: [640]},
                     "C": {3: [640, 1280, 1280], 2: [640, 1280], 1: [1280]}, "D": {3: [1280] * 3, 2: [1280] * 2, 1: [1280]}}


class ControlNetAdapterOracle(nn.Module):
    """ControlNetAdapter (model/ctrl_adapter.py:17-224), num_repeats == 1 (the shipped configurations)."""

    def __init__(self, backbone_model_name, num_blocks=2, num_frames=8, num_adapters_per_location=3,
                 cross_attention_dim=None, add_spatial_resnet=True, add_temporal_resnet=False,
                 add_spatial_transformer=True, add_temporal_transformer=False, add_adapter_location_A=False,
                 add_adapter_location_B=False, add_adapter_location_C=False, add_adapter_location_D=False,
                 add_adapter_location_M=False):
        super().__init__()
        locs = [k for k, on in zip("ABCD", (add_adapter_location_A, add_adapter_location_B, add_adapter_location_C,
                                            add_adapter_location_D)) if on]
        n = num_adapters_per_location
        self.slot_ids = sum((LOCATION_SLOTS[k][n] for k in locs), [])
        chans = sum((LOCATION_CHANNELS[k][n] for k in locs), [])
        up = 2 if backbone_model_name in ["sdxl"] else 1
        kw = dict(cross_attention_dim=cross_attention_dim, num_layers=num_blocks, up_sampling_scale=up,
                  add_spatial_resnet=add_spatial_resnet, add_temporal_resnet=add_temporal_resnet,
                  add_spatial_transformer=add_spatial_transformer, add_temporal_transformer=add_temporal_transformer)
        self.down_blocks_adapter = nn.ModuleList([AdapterBlockOracle(cc, cc, **kw) for cc in chans])
        self.mid_block_adapter = AdapterBlockOracle(1280, 1280, **kw) if add_adapter_location_M else None

    def forward(self, down_block_res_samples, mid_block_res_sample=None, sparsity_masking=None, num_frames=None,
                timestep=None, encoder_hidden_states=None):
        out, k = [], 0
        for i in range(12):
            if i in self.slot_ids:
                out.append(self.down_blocks_adapter[k](down_block_res_samples[i], num_frames=num_frames,
                                                       timestep=timestep, encoder_hidden_states=encoder_hidden_states))
                k += 1
            else:
                out.append(torch.zeros_like(down_block_res_samples[i]))
        mid = None
        if mid_block_res_sample is not None and self.mid_block_adapter is not None:
            mid = self.mid_block_adapter(mid_block_res_sample, num_frames=num_frames, timestep=timestep,
                                         encoder_hidden_states=encoder_hidden_states)
        return out, mid
