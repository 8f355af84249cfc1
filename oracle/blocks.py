"""diffusers v0.27.x building blocks, restated in plain PyTorch (TEST INFRASTRUCTURE -- see oracle/__init__.py).

`diffusers` is an un-vendored, unpinned dependency of the reference (requirements_inference.txt:1); the
reference targets v0.27.2 (sdxl/pipelines/sdxl_controlnet_adapter_pipeline.py:2, model/resnet_block_2d.py:28
cites commit 2dcf64b7).  Each class below restates the published algorithm of the named diffusers file and
keeps the attribute names that determine the state-dict keys.  PARITY: unpinned against diffusers itself (not
installable here); cross-checked against independent naive formulations in tests/test_oracle_blocks.py.

Call sites in the reference that fix how these are used:
  model/adapter_spatial_temporal.py:7,96,108,120,134 ; model/resnet_block_2d.py:8-25 ;
  controlnet/controlnet.py:16-35,371-424
"""
import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn


# ---- diffusers/models/embeddings.py -------------------------------------------------------------------
def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1.0,
                           scale=1.0, max_period=10000):
    assert timesteps.dim() == 1
    half_dim = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half_dim, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half_dim - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half_dim:], emb[:, :half_dim]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, flip_sin_to_cos=self.flip_sin_to_cos,
                                      downscale_freq_shift=self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim if out_dim is not None else time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


# ---- diffusers/models/attention_processor.py (Attention + AttnProcessor2_0) -----------------------------
class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False, out_bias=True):
        super().__init__()
        self.inner_dim = dim_head * heads
        self.heads = heads
        self.scale = dim_head ** -0.5
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(kv_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(kv_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(0.0)])

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None):
        b, lq, _ = hidden_states.shape
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q, k, v = self.to_q(hidden_states), self.to_k(ctx), self.to_v(ctx)
        hd = self.inner_dim // self.heads
        q = q.view(b, -1, self.heads, hd).transpose(1, 2)
        k = k.view(b, -1, self.heads, hd).transpose(1, 2)
        v = v.view(b, -1, self.heads, hd).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(b, -1, self.heads * hd)
        return self.to_out[1](self.to_out[0](o))


# ---- diffusers/models/attention.py ------------------------------------------------------------------
class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)          # exact (erf) GELU


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, activation_fn="geglu"):
        super().__init__()
        inner = int(dim * mult)
        dim_out = dim_out if dim_out is not None else dim
        assert activation_fn == "geglu"
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(0.0), nn.Linear(inner, dim_out)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    """layer_norm variant, attention_bias=False, activation geglu, dropout 0 (the only one reached)."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, cross_attention_dim=None):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, None, num_attention_heads, attention_head_dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, cross_attention_dim, num_attention_heads, attention_head_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, **kw):
        hidden_states = self.attn1(self.norm1(hidden_states)) + hidden_states
        hidden_states = self.attn2(self.norm2(hidden_states), encoder_hidden_states=encoder_hidden_states) + hidden_states
        hidden_states = self.ff(self.norm3(hidden_states)) + hidden_states
        return hidden_states


class TemporalBasicTransformerBlock(nn.Module):
    def __init__(self, dim, time_mix_inner_dim, num_attention_heads, attention_head_dim, cross_attention_dim=None):
        super().__init__()
        self.is_res = dim == time_mix_inner_dim
        self.norm_in = nn.LayerNorm(dim)
        self.ff_in = FeedForward(dim, dim_out=time_mix_inner_dim)
        self.norm1 = nn.LayerNorm(time_mix_inner_dim)
        self.attn1 = Attention(time_mix_inner_dim, None, num_attention_heads, attention_head_dim)
        if cross_attention_dim is not None:
            self.norm2 = nn.LayerNorm(time_mix_inner_dim)
            self.attn2 = Attention(time_mix_inner_dim, cross_attention_dim, num_attention_heads, attention_head_dim)
        else:
            self.norm2 = None
            self.attn2 = None
        self.norm3 = nn.LayerNorm(time_mix_inner_dim)
        self.ff = FeedForward(time_mix_inner_dim)

    def forward(self, hidden_states, num_frames, encoder_hidden_states=None):
        batch_frames, seq_length, channels = hidden_states.shape
        batch_size = batch_frames // num_frames
        hidden_states = hidden_states[None, :].reshape(batch_size, num_frames, seq_length, channels)
        hidden_states = hidden_states.permute(0, 2, 1, 3)
        hidden_states = hidden_states.reshape(batch_size * seq_length, num_frames, channels)
        residual = hidden_states
        hidden_states = self.ff_in(self.norm_in(hidden_states))
        if self.is_res:
            hidden_states = hidden_states + residual
        hidden_states = self.attn1(self.norm1(hidden_states), encoder_hidden_states=None) + hidden_states
        if self.attn2 is not None:
            hidden_states = self.attn2(self.norm2(hidden_states), encoder_hidden_states=encoder_hidden_states) + hidden_states
        ff_output = self.ff(self.norm3(hidden_states))
        hidden_states = ff_output + hidden_states if self.is_res else ff_output
        hidden_states = hidden_states[None, :].reshape(batch_size, seq_length, num_frames, channels)
        hidden_states = hidden_states.permute(0, 2, 1, 3)
        hidden_states = hidden_states.reshape(batch_size * num_frames, seq_length, channels)
        return hidden_states


# ---- diffusers/models/upsampling.py, downsampling.py ----------------------------------------------------
class Upsample2D(nn.Module):
    def __init__(self, channels, use_conv=False):
        super().__init__()
        assert not use_conv
        self.channels = channels

    def forward(self, hidden_states, output_size=None):
        assert hidden_states.shape[1] == self.channels
        if output_size is None:
            return F.interpolate(hidden_states, scale_factor=2.0, mode="nearest")
        return F.interpolate(hidden_states, size=output_size, mode="nearest")


class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=True, out_channels=None, padding=1, name="conv"):
        super().__init__()
        assert use_conv
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=padding)

    def forward(self, hidden_states):
        return self.conv(hidden_states)


# ---- diffusers/models/resnet.py ---------------------------------------------------------------------
class ResnetBlock2D(nn.Module):
    """default time_embedding_norm, swish, groups 32 (the configuration get_down_block builds)."""

    def __init__(self, *, in_channels, out_channels=None, temb_channels=512, groups=32, eps=1e-6,
                 output_scale_factor=1.0, use_in_shortcut=None):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.output_scale_factor = output_scale_factor
        use_in_shortcut = (in_channels != out_channels) if use_in_shortcut is None else use_in_shortcut
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if use_in_shortcut else None

    def forward(self, input_tensor, temb):
        h = self.conv1(F.silu(self.norm1(input_tensor)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(self.dropout(F.silu(self.norm2(h))))
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + h) / self.output_scale_factor


class TemporalResnetBlock(nn.Module):
    def __init__(self, in_channels, out_channels=None, temb_channels=512, eps=1e-6):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.norm1 = nn.GroupNorm(32, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv3d(in_channels, out_channels, (3, 1, 1), stride=1, padding=(1, 0, 0))
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(32, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv3d(out_channels, out_channels, (3, 1, 1), stride=1, padding=(1, 0, 0))
        self.conv_shortcut = None
        if in_channels != out_channels:
            self.conv_shortcut = nn.Conv3d(in_channels, out_channels, 1)

    def forward(self, input_tensor, temb):
        h = self.conv1(F.silu(self.norm1(input_tensor)))
        if self.time_emb_proj is not None:
            t = self.time_emb_proj(F.silu(temb))[:, :, :, None, None]
            h = h + t.permute(0, 2, 1, 3, 4)
        h = self.conv2(self.dropout(F.silu(self.norm2(h))))
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return input_tensor + h


class AlphaBlender(nn.Module):
    def __init__(self, alpha, merge_strategy="learned_with_images", switch_spatial_to_temporal_mix=False):
        super().__init__()
        self.merge_strategy = merge_strategy
        self.switch_spatial_to_temporal_mix = switch_spatial_to_temporal_mix
        assert merge_strategy in ("learned", "learned_with_images")
        self.register_parameter("mix_factor", nn.Parameter(torch.Tensor([alpha])))

    def get_alpha(self, image_only_indicator, ndims):
        if self.merge_strategy == "learned":
            alpha = torch.sigmoid(self.mix_factor)
        else:
            alpha = torch.where(image_only_indicator.bool(),
                                torch.ones(1, 1, device=image_only_indicator.device),
                                torch.sigmoid(self.mix_factor)[..., None])
            if ndims == 5:
                alpha = alpha[:, None, :, None, None]
            elif ndims == 3:
                alpha = alpha.reshape(-1)[:, None, None]
            else:
                raise ValueError(ndims)
        return alpha

    def forward(self, x_spatial, x_temporal, image_only_indicator=None):
        alpha = self.get_alpha(image_only_indicator, x_spatial.ndim).to(x_spatial.dtype)
        if self.switch_spatial_to_temporal_mix:
            alpha = 1.0 - alpha
        return alpha * x_spatial + (1.0 - alpha) * x_temporal


# ---- diffusers/models/transformers/transformer_2d.py (conv projections, 1 layer) --------------------------
class Transformer2DModel(nn.Module):
    def __init__(self, num_attention_heads, attention_head_dim, in_channels, num_layers=1, cross_attention_dim=None,
                 norm_num_groups=32):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim)
            for _ in range(num_layers)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)

    def forward(self, hidden_states, encoder_hidden_states=None):
        b, _, h, w = hidden_states.shape
        residual = hidden_states
        x = self.proj_in(self.norm(hidden_states))
        inner = x.shape[1]
        x = x.permute(0, 2, 3, 1).reshape(b, h * w, inner)
        for blk in self.transformer_blocks:
            x = blk(x, encoder_hidden_states=encoder_hidden_states)
        x = x.reshape(b, h, w, inner).permute(0, 3, 1, 2).contiguous()
        return self.proj_out(x) + residual


# ---- diffusers/models/unets/unet_2d_blocks.py ----------------------------------------------------------
class CrossAttnDownBlock2D(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, resnet_eps, num_attention_heads,
                 cross_attention_dim, add_downsample, downsample_padding=1, resnet_groups=32):
        super().__init__()
        self.resnets = nn.ModuleList()
        self.attentions = nn.ModuleList()
        for i in range(num_layers):
            cin = in_channels if i == 0 else out_channels
            self.resnets.append(ResnetBlock2D(in_channels=cin, out_channels=out_channels, temb_channels=temb_channels,
                                              eps=resnet_eps, groups=resnet_groups))
            self.attentions.append(Transformer2DModel(num_attention_heads, out_channels // num_attention_heads,
                                                      out_channels, 1, cross_attention_dim, resnet_groups))
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, True, out_channels, downsample_padding, "op")]) \
            if add_downsample else None

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, **kw):
        outs = ()
        for resnet, attn in zip(self.resnets, self.attentions):
            hidden_states = attn(resnet(hidden_states, temb), encoder_hidden_states=encoder_hidden_states)
            outs = outs + (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            outs = outs + (hidden_states,)
        return hidden_states, outs


class DownBlock2D(nn.Module):
    has_cross_attention = False

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, resnet_eps, add_downsample,
                 downsample_padding=1, resnet_groups=32):
        super().__init__()
        self.resnets = nn.ModuleList()
        for i in range(num_layers):
            cin = in_channels if i == 0 else out_channels
            self.resnets.append(ResnetBlock2D(in_channels=cin, out_channels=out_channels, temb_channels=temb_channels,
                                              eps=resnet_eps, groups=resnet_groups))
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels, True, out_channels, downsample_padding, "op")]) \
            if add_downsample else None

    def forward(self, hidden_states, temb=None, **kw):
        outs = ()
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states, temb)
            outs = outs + (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            outs = outs + (hidden_states,)
        return hidden_states, outs


def get_down_block(down_block_type, num_layers, in_channels, out_channels, temb_channels, add_downsample, resnet_eps,
                   resnet_act_fn="silu", transformer_layers_per_block=1, num_attention_heads=None, resnet_groups=32,
                   cross_attention_dim=None, downsample_padding=1, **unused):
    assert resnet_act_fn in ("silu", "swish") and transformer_layers_per_block == 1
    if down_block_type == "CrossAttnDownBlock2D":
        return CrossAttnDownBlock2D(in_channels, out_channels, temb_channels, num_layers, resnet_eps,
                                    num_attention_heads, cross_attention_dim, add_downsample, downsample_padding,
                                    resnet_groups)
    if down_block_type == "DownBlock2D":
        return DownBlock2D(in_channels, out_channels, temb_channels, num_layers, resnet_eps, add_downsample,
                           downsample_padding, resnet_groups)
    raise ValueError(down_block_type)


class UNetMidBlock2DCrossAttn(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, temb_channels, resnet_eps=1e-6, num_attention_heads=1, cross_attention_dim=1280,
                 resnet_groups=32, output_scale_factor=1.0, transformer_layers_per_block=1, **unused):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels=in_channels, out_channels=in_channels,
                                                    temb_channels=temb_channels, eps=resnet_eps, groups=resnet_groups,
                                                    output_scale_factor=output_scale_factor)])
        self.attentions = nn.ModuleList([Transformer2DModel(num_attention_heads, in_channels // num_attention_heads,
                                                            in_channels, 1, cross_attention_dim, resnet_groups)])
        self.resnets.append(ResnetBlock2D(in_channels=in_channels, out_channels=in_channels,
                                          temb_channels=temb_channels, eps=resnet_eps, groups=resnet_groups,
                                          output_scale_factor=output_scale_factor))

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, **kw):
        hidden_states = self.resnets[0](hidden_states, temb)
        hidden_states = self.attentions[0](hidden_states, encoder_hidden_states=encoder_hidden_states)
        return self.resnets[1](hidden_states, temb)
