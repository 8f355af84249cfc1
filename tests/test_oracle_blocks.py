"""Self-consistency of the diffusers-block restatements (oracle/blocks.py) against independent naive
formulations -- the stand-in for the un-installable dependency.  CPU only."""
import math

import torch
import torch.nn.functional as F

from helpers import ROOT  # noqa: F401
from oracle import blocks as B
from oracle.init import seeded_init, seeded_tensor

torch.set_grad_enabled(False)


def test_timestep_embedding_formula():
    t = torch.tensor([0.0, 1.0, 999.0])
    e = B.Timesteps(320, True, 0)(t)
    half = 160
    for n in range(3):
        for k in (0, 1, 77, 159):
            f = math.exp(-math.log(10000.0) * k / half)
            assert abs(e[n, k].item() - math.cos(t[n].item() * f)) < 1e-4          # [cos | sin] (flip_sin_to_cos)
            assert abs(e[n, half + k].item() - math.sin(t[n].item() * f)) < 1e-4


def test_attention_matches_explicit_softmax():
    a = seeded_init(B.Attention(64, None, heads=4, dim_head=8), 1)
    x = seeded_tensor((2, 10, 64), 3)
    q, k, v = a.to_q(x), a.to_k(x), a.to_v(x)
    out = torch.zeros(2, 10, 32)
    for b in range(2):
        for h in range(4):
            sl = slice(8 * h, 8 * h + 8)
            s = q[b, :, sl] @ k[b, :, sl].t() / math.sqrt(8)
            p = torch.exp(s - s.max(-1, keepdim=True).values)
            p = p / p.sum(-1, keepdim=True)
            out[b, :, sl] = p @ v[b, :, sl]
    ref = out @ a.to_out[0].weight.t() + a.to_out[0].bias
    assert torch.allclose(a(x), ref, atol=1e-5)
    # cross-attention with ONE key is query-independent (SURVEY.md note N5)
    c = seeded_init(B.Attention(64, 24, heads=4, dim_head=8), 2)
    ctx = seeded_tensor((2, 1, 24), 4)
    y = c(x, encoder_hidden_states=ctx)
    const = c.to_out[0](c.to_v(ctx))
    assert torch.allclose(y, const.expand_as(y), atol=1e-5)


def test_geglu_and_groupnorm_layernorm_formulas():
    ff = seeded_init(B.FeedForward(32), 5)
    x = seeded_tensor((3, 7, 32), 6)
    y = ff.net[0].proj(x)
    h, g = y[..., :128], y[..., 128:]
    gelu = 0.5 * g * (1 + torch.erf(g / math.sqrt(2)))
    assert torch.allclose(ff(x), ff.net[2](h * gelu), atol=1e-5)
    gn = seeded_init(torch.nn.GroupNorm(4, 16, eps=1e-6), 7)
    z = seeded_tensor((2, 16, 5, 5), 8)
    zz = z.reshape(2, 4, -1)
    ref = ((zz - zz.mean(-1, keepdim=True)) / torch.sqrt(zz.var(-1, unbiased=False, keepdim=True) + 1e-6)).reshape(z.shape)
    ref = ref * gn.weight[None, :, None, None] + gn.bias[None, :, None, None]
    assert torch.allclose(gn(z), ref, atol=1e-5)


def test_conv_via_unfold_and_upsample():
    r = seeded_init(B.ResnetBlock2D(in_channels=32, out_channels=64, temb_channels=16, eps=1e-5), 9)
    x = seeded_tensor((2, 32, 6, 6), 10)
    cols = F.unfold(F.silu(r.norm1(x)), 3, padding=1)                 # explicit im2col
    y = (r.conv1.weight.reshape(64, -1) @ cols).reshape(2, 64, 6, 6) + r.conv1.bias[None, :, None, None]
    assert torch.allclose(r.conv1(F.silu(r.norm1(x))), y, atol=1e-4)
    up = B.Upsample2D(32)(x, (12, 12))
    assert torch.equal(up[:, :, ::2, ::2], x) and torch.equal(up[:, :, 1::2, 1::2], x)


def test_temporal_block_is_pointwise_in_space_and_blender():
    blk = seeded_init(B.TemporalBasicTransformerBlock(32, 32, 2, 16, cross_attention_dim=8), 11)
    x = seeded_tensor((2 * 3, 5, 32), 12)          # b=2, F=3, L=5
    ctx = seeded_tensor((1, 1, 8), 13).expand(2 * 5, 1, 8)
    y = blk(x, num_frames=3, encoder_hidden_states=ctx)
    # running one pixel alone gives the same result (no mixing across the L axis)
    y1 = blk(x[:, 2:3], num_frames=3, encoder_hidden_states=ctx[:2])
    assert torch.allclose(y[:, 2:3], y1, atol=1e-5)
    ab = B.AlphaBlender(0.5)
    xs, xt = seeded_tensor((2, 4, 3, 2, 2), 14), seeded_tensor((2, 4, 3, 2, 2), 15)
    a = torch.sigmoid(torch.tensor(0.5))
    assert torch.allclose(ab(xs, xt, torch.zeros(2, 3)), a * xs + (1 - a) * xt, atol=1e-6)
    tr = seeded_init(B.TemporalResnetBlock(32, 32, 16, eps=1e-6), 16)
    v = seeded_tensor((1, 32, 4, 3, 3), 17)
    out = tr(v, seeded_tensor((1, 4, 16), 18))
    assert out.shape == v.shape


def test_attention_matches_pytorch_multihead_attention():
    """an INDEPENDENT implementation of the same operator (PyTorch core's nn.MultiheadAttention: q/k/v projections without
    bias, 1/sqrt(head_dim) scaling, softmax, output projection with bias) against the restatement of diffusers' Attention,
    self-attention and cross-attention with a different context width (kdim / vdim)"""
    for cross in (None, 24):
        a = seeded_init(B.Attention(64, cross, heads=4, dim_head=16), 21)
        mha = torch.nn.MultiheadAttention(64, 4, bias=True, batch_first=True, kdim=cross, vdim=cross)
        with torch.no_grad():
            if cross is None:
                mha.in_proj_weight.copy_(torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight]))
            else:
                mha.q_proj_weight.copy_(a.to_q.weight); mha.k_proj_weight.copy_(a.to_k.weight); mha.v_proj_weight.copy_(a.to_v.weight)
            mha.in_proj_bias.zero_()
            mha.out_proj.weight.copy_(a.to_out[0].weight); mha.out_proj.bias.copy_(a.to_out[0].bias)
        x = seeded_tensor((3, 11, 64), 22)
        ctx = x if cross is None else seeded_tensor((3, 7, cross), 23)
        ref, _ = mha(x, ctx, ctx, need_weights=False)
        got = a(x, encoder_hidden_states=None if cross is None else ctx)
        assert torch.allclose(got, ref, atol=2e-5), (cross, (got - ref).abs().max())


def test_basic_transformer_block_matches_pytorch_prenorm_decoder_layer():
    """structure check against PyTorch core's pre-norm nn.TransformerDecoderLayer (self-attention, cross-attention,
    feed-forward, each `x + f(norm(x))`), with the feed-forward activation replaced by the GEGLU gate.  The layer's
    cross-attention has no kdim, so the context width equals the model width here."""
    d, heads, inner = 64, 4, 256
    blk = seeded_init(B.BasicTransformerBlock(d, heads, d // heads, cross_attention_dim=d), 31)

    def geglu(h):
        a, g = h.chunk(2, dim=-1)
        return a * torch.nn.functional.gelu(g)
    ref = torch.nn.TransformerDecoderLayer(d, heads, dim_feedforward=2 * inner, dropout=0.0, activation=geglu,
                                           layer_norm_eps=1e-5, batch_first=True, norm_first=True)
    ref.linear2 = torch.nn.Linear(inner, d)
    with torch.no_grad():
        for mha, a in ((ref.self_attn, blk.attn1), (ref.multihead_attn, blk.attn2)):
            mha.in_proj_weight.copy_(torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight]))
            mha.in_proj_bias.zero_()
            mha.out_proj.weight.copy_(a.to_out[0].weight); mha.out_proj.bias.copy_(a.to_out[0].bias)
        for n_ref, n in ((ref.norm1, blk.norm1), (ref.norm2, blk.norm2), (ref.norm3, blk.norm3)):
            n_ref.weight.copy_(n.weight); n_ref.bias.copy_(n.bias)
        ref.linear1.weight.copy_(blk.ff.net[0].proj.weight); ref.linear1.bias.copy_(blk.ff.net[0].proj.bias)
        ref.linear2.weight.copy_(blk.ff.net[2].weight); ref.linear2.bias.copy_(blk.ff.net[2].bias)
    ref.train()          # keeps PyTorch on the plain (non-fused) path; dropout is 0
    x, ctx = seeded_tensor((2, 9, d), 32), seeded_tensor((2, 5, d), 33)
    want = ref(x, ctx)
    got = blk(x, encoder_hidden_states=ctx)
    assert torch.allclose(got, want, atol=5e-5), (got - want).abs().max()
