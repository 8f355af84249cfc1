"""Op-level parity: every hand-written HIP kernel (through the ctrl_op_* C ABI) against a plain PyTorch
fp32 reference of the same op on the CPU.  Inputs are rounded to fp16 first so both sides see identical
values; the tolerance (stated per test) covers fp16 output rounding (2^-11) plus fp32 accumulation order.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_inf

pytestmark = pytest.mark.gpu

TOL = 2e-3   # rel-inf; one fp16 rounding of the result is 4.9e-4 relative


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half().float()


@pytest.fixture(scope="module")
def ops(gpu):
    import ctrl_adapter_amd  # noqa: F401
    from ctrl_adapter_amd import ops as o
    return o


def report(name, err, tol=TOL):
    print("PARITY %-40s rel_inf=%.3e (tol %.1e)" % (name, err, tol))
    assert err <= tol, "%s: rel_inf %.3e > %.1e" % (name, err, tol)


@pytest.mark.parametrize("M,N,K", [(300, 320, 512), (4096, 1280, 1280), (1000, 960, 512), (77 * 2, 2560, 2048), (130, 512, 320)])
def test_linear(ops, gpu, M, N, K):
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3), rnd(M, N, seed=4)
    ref = x @ w.t() + b + r
    wp = ops.pack_linear_w(w.to(gpu))
    out = ops.linear(x.half().to(gpu), wp, bias=b.to(gpu), res=r.half().to(gpu))
    report("linear %dx%dx%d" % (M, N, K), rel_inf(out, ref))


def test_linear_f32_stream(ops, gpu):
    """fp32 residual stream: fp32 residual in, fp32 master out + fp16 operand mirror (ctrl_igemm_desc.res_f32 / out16)"""
    M, N, K = 700, 512, 320
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3)
    r = torch.randn(M, N, generator=torch.Generator().manual_seed(4))          # NOT fp16-representable
    ref = x @ w.t() + b + r
    wp = ops.pack_linear_w(w.to(gpu))
    out = torch.empty(M, N, dtype=torch.float32, device=gpu)
    mirror = torch.empty(M, N, dtype=torch.float16, device=gpu)
    ops.igemm(x.half().to(gpu), K, wp, M, N, K, bias=b.to(gpu), res=r.to(gpu), ldres=N,
              segs=[(out, N, 0, N, ops.SEG_ROW, 1)], out16=mirror, ld16=N)
    report("linear f32 stream master", rel_inf(out, ref), 2e-5)
    report("linear f32 stream mirror", rel_inf(mirror, ref))


def test_linear_geglu(ops, gpu):
    M, K, inner = 260, 512, 2048
    x, w, b = rnd(M, K, seed=1), rnd(2 * inner, K, seed=2, scale=0.05), rnd(2 * inner, seed=3)
    y = x @ w.t() + b
    ref = y[:, :inner] * F.gelu(y[:, inner:])
    wp = ops.pack_linear_w(w.to(gpu), geglu=True)
    bp = ops.pack_vec(b.to(gpu), geglu=True)
    out = ops.linear(x.half().to(gpu), wp, bias=bp, geglu=True)
    report("linear_geglu", rel_inf(out, ref))


@pytest.mark.parametrize("cin,cout,h,stride,up", [(320, 320, 16, 1, 1), (320, 640, 16, 2, 1), (320, 320, 8, 1, 2),
                                                  (96, 96, 32, 1, 1), (96, 256, 32, 2, 1), (640, 640, 32, 1, 1)])
def test_conv3x3(ops, gpu, cin, cout, h, stride, up):
    n = 2
    x = rnd(n, cin, h, h, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2, scale=0.02)
    b = rnd(cout, seed=3)
    temb = rnd(n, cout, seed=4)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up == 2 else x
    ref = F.conv2d(xin, w, b, stride=stride, padding=1) + temb[:, :, None, None]
    wp = ops.pack_conv_w(w.to(gpu))
    xh = x.permute(0, 2, 3, 1).contiguous().half().to(gpu)
    out = ops.conv2d(xh, wp, cout, taps=9, stride=stride, up=up, bias=b.to(gpu), rowvec=temb.to(gpu))
    report("conv3x3 %d->%d s%d up%d" % (cin, cout, stride, up), rel_inf(out.permute(0, 3, 1, 2), ref))


@pytest.mark.parametrize("cin,cout,h,n", [(1280, 1280, 8, 8), (640, 640, 16, 8), (1280, 1280, 16, 2)])
def test_conv3x3_splitk(ops, gpu, cin, cout, h, n):
    """small-M / long-K convs (ControlNet low-resolution blocks) take the split-K path when scratch is provided"""
    x = rnd(n, cin, h, h, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2, scale=0.02)
    b, temb, r = rnd(cout, seed=3), rnd(n, cout, seed=4), rnd(n, cout, h, h, seed=5)
    ref = F.conv2d(x, w, b, padding=1) + temb[:, :, None, None] + r
    wp = ops.pack_conv_w(w.to(gpu))
    xh = x.permute(0, 2, 3, 1).contiguous().half().to(gpu)
    rh = r.permute(0, 2, 3, 1).contiguous().half().to(gpu)
    ws = torch.empty(16 * n * h * h * cout, dtype=torch.float32, device=gpu)
    out = ops.conv2d(xh, wp, cout, taps=9, bias=b.to(gpu), rowvec=temb.to(gpu), res=rh, splitk_ws=ws)
    report("conv3x3 split-K %d->%d @%d n%d" % (cin, cout, h, n), rel_inf(out.permute(0, 3, 1, 2), ref))
    out2 = ops.conv2d(xh, wp, cout, taps=9, bias=b.to(gpu), rowvec=temb.to(gpu), res=rh)       # same problem, no split
    report("conv3x3 no-split %d->%d @%d n%d" % (cin, cout, h, n), rel_inf(out2.permute(0, 3, 1, 2), ref))


def test_conv1x1_nchw_out_scaled(ops, gpu):
    n, cin, cout, h = 2, 320, 320, 16
    x, w, b = rnd(n, cin, h, h, seed=1), rnd(cout, cin, 1, 1, seed=2, scale=0.05), rnd(cout, seed=3)
    ref = F.conv2d(x, w, b) * 0.7
    wp = ops.pack_conv_w(w.to(gpu))
    xh = x.permute(0, 2, 3, 1).contiguous().half().to(gpu)
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        out = ops.conv2d(xh, wp, cout, taps=1, bias=b.to(gpu), scale=0.7, out_nchw_dtype=dt)
        report("conv1x1 nchw %s" % dt, rel_inf(out, ref), 1e-2 if dt == torch.bfloat16 else TOL)


def test_conv1x1_upsampled_shortcut(ops, gpu):
    n, c, h = 2, 320, 8
    x, w, b = rnd(n, c, h, h, seed=1), rnd(c, c, 1, 1, seed=2, scale=0.05), rnd(c, seed=3)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b)
    wp = ops.pack_conv_w(w.to(gpu))
    xh = x.permute(0, 2, 3, 1).contiguous().half().to(gpu)
    out = ops.conv2d(xh, wp, c, taps=1, up=2, bias=b.to(gpu))
    report("conv1x1 up2", rel_inf(out.permute(0, 3, 1, 2), ref))


@pytest.mark.parametrize("c,h,n,splitk", [(320, 16, 2, False), (320, 32, 8, True), (640, 8, 3, False)])
def test_conv3x3_with_half_resolution_residual(ops, gpu, c, h, n, splitk):
    """ctrl_igemm_desc::res_up = 2 -- the adapter's ResnetBlock2D with up-sampling (model/resnet_block_2d.py:174-184,216): the 1x1
    shortcut conv commutes with the nearest x2 up-sampling in front of it, so it runs on the input grid and conv2's epilogue
    (and the split-K finish kernel) reads it as res[n][oy/2][ox/2].  Bit-identical to the full-resolution residual."""
    x = rnd(n, c, 2 * h, 2 * h, seed=1)                 # conv2's operand (already at the up-sampled resolution)
    w = rnd(c, c, 3, 3, seed=2, scale=0.02)
    b = rnd(c, seed=3)
    r_low = torch.randn(n, h, h, c, generator=torch.Generator().manual_seed(4))      # the shortcut's fp32 output, input grid
    wp = ops.pack_conv_w(w.to(gpu))
    xh = x.permute(0, 2, 3, 1).contiguous().half().to(gpu)
    r_full = r_low.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2).contiguous()
    ws = torch.empty(16 * n * 4 * h * h * c, dtype=torch.float32, device=gpu) if splitk else None
    got = ops.conv2d(xh, wp, c, taps=9, bias=b.to(gpu), res=r_low.to(gpu), res_up=2, splitk_ws=ws)
    want = ops.conv2d(xh, wp, c, taps=9, bias=b.to(gpu), res=r_full.to(gpu), splitk_ws=ws)
    assert torch.equal(got, want)
    ref = F.conv2d(x.half().float(), w.half().float(), b, padding=1) + r_full.permute(0, 3, 1, 2)
    report("conv3x3 + up-sampled residual c%d h%d" % (c, 2 * h), rel_inf(got.permute(0, 3, 1, 2), ref))


def test_qkv_segments(ops, gpu):
    B, Ltok, K, Cc = 2, 192, 512, 320
    x = rnd(B * Ltok, K, seed=1)
    w = rnd(3 * Cc, K, seed=2, scale=0.05)
    ref = x @ w.t()
    wp = ops.pack_linear_w(w.to(gpu))
    qk = torch.zeros(B * Ltok, 2 * Cc, dtype=torch.float16, device=gpu)
    Lpad = 256
    vt = torch.zeros(B, Cc, Lpad, dtype=torch.float16, device=gpu)
    ops.igemm(x.half().to(gpu), K, wp, B * Ltok, 3 * Cc, K,
              segs=[(qk, 2 * Cc, 0, 2 * Cc, ops.SEG_ROW, 1), (vt, Lpad, 2 * Cc, Cc, ops.SEG_TRANSPOSED, Ltok)])
    report("qkv seg q|k", rel_inf(qk, ref[:, :2 * Cc]))
    vref = ref[:, 2 * Cc:].reshape(B, Ltok, Cc).permute(0, 2, 1)
    report("qkv seg v^T", rel_inf(vt[:, :, :Ltok], vref))
    assert vt[:, :, Ltok:].abs().max().item() == 0.0
    # odd token count (77) exercises the scalar transposed store
    B2, L2 = 3, 77
    x2 = rnd(B2 * L2, K, seed=5)
    ref2 = x2 @ w.t()
    qk2 = torch.zeros(B2 * L2, 2 * Cc, dtype=torch.float16, device=gpu)
    vt2 = torch.zeros(B2, Cc, 128, dtype=torch.float16, device=gpu)
    ops.igemm(x2.half().to(gpu), K, wp, B2 * L2, 3 * Cc, K,
              segs=[(qk2, 2 * Cc, 0, 2 * Cc, ops.SEG_ROW, 1), (vt2, 128, 2 * Cc, Cc, ops.SEG_TRANSPOSED, L2)])
    report("qkv seg (L=77) q|k", rel_inf(qk2, ref2[:, :2 * Cc]))
    report("qkv seg (L=77) v^T", rel_inf(vt2[:, :, :L2], ref2[:, 2 * Cc:].reshape(B2, L2, Cc).permute(0, 2, 1)))


@pytest.mark.parametrize("B,Ltok,K,N", [(3, 16384, 512, 320), (2, 64, 1280, 1280), (5, 192, 320, 640), (8, 4096, 512, 640)])
def test_single_transposed_output_epilogue(ops, gpu, B, Ltok, K, N):
    """proj_out -> NCHW and the V^T operand: one transposed segment goes through the LDS-transposed epilogue of the
    swapped-operand kernel; bias, per-image vector, residual (fp32 / fp16, row-major), scale and the three output
    dtypes; token counts that put two images into one wave tile (64) or leave an M tail (5 x 192)."""
    M = B * Ltok
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    bias, rv, res = rnd(N, seed=3), rnd(B, N, seed=4), rnd(M, N, seed=5)
    wp = ops.pack_linear_w(w.to(gpu))
    xg = x.half().to(gpu)
    Lpad = (Ltok + 63) // 64 * 64 + 64
    base = x.half().float() @ w.half().float().t() + bias + rv.repeat_interleave(Ltok, 0)
    for res_dt, out_dt in ((torch.float32, torch.float16), (torch.float16, torch.float32), (None, torch.bfloat16)):
        r = res.to(res_dt).to(gpu) if res_dt is not None else None
        ref = (base + (r.float().cpu() if r is not None else 0)) * 0.7
        out = torch.zeros(B, N, Lpad, dtype=out_dt, device=gpu)
        ops.igemm(xg, K, wp, M, N, K, bias=bias.to(gpu), rowvec=rv.to(gpu), rows_per_img=Ltok, res=r, ldres=N, scale=0.7,
                  segs=[(out, Lpad, 0, N, ops.SEG_TRANSPOSED, Ltok)])
        report("transposed out B%d L%d K%d N%d res=%s out=%s" % (B, Ltok, K, N, res_dt, out_dt),
               rel_inf(out[:, :, :Ltok], ref.reshape(B, Ltok, N).permute(0, 2, 1)), 1e-2 if out_dt == torch.bfloat16 else TOL)
        assert out[:, :, Ltok:].abs().max().item() == 0.0


def test_temporal_conv(ops, gpu):
    b, Fr, HW, Cc = 2, 5, 48, 64
    x = rnd(b, Cc, Fr, HW, 1, seed=1)            # b c f h w
    w = rnd(Cc, Cc, 3, 1, 1, seed=2, scale=0.05)
    bias = rnd(Cc, seed=3)
    ref = F.conv3d(x, w, bias, padding=(1, 0, 0))  # b c f hw 1
    xr = x[..., 0].permute(0, 2, 3, 1).contiguous().reshape(b * Fr * HW, Cc)   # [(b f hw)][c]
    wp = ops.pack_conv_w(w.reshape(Cc, Cc, 3, 1).to(gpu))
    out = torch.empty(b * Fr * HW, Cc, dtype=torch.float16, device=gpu)
    ops.igemm(xr.half().to(gpu), Cc, wp, b * Fr * HW, Cc, Cc, taps=3, mode=ops.IG_TEMPORAL, bias=bias.to(gpu),
              segs=[(out, Cc, 0, Cc, ops.SEG_ROW, 1)], F=Fr, HW=HW)
    refr = ref[..., 0].permute(0, 2, 3, 1).reshape(b * Fr * HW, Cc)
    report("temporal conv3", rel_inf(out, refr))


def _attn_ref(q, k, v, heads):
    B, Lq, Cc = q.shape
    D = Cc // heads
    qh = q.reshape(B, Lq, heads, D).transpose(1, 2)
    kh = k.reshape(B, -1, heads, D).transpose(1, 2)
    vh = v.reshape(B, -1, heads, D).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) / math.sqrt(D)
    o = torch.softmax(s, dim=-1) @ vh
    return o.transpose(1, 2).reshape(B, Lq, Cc)


@pytest.mark.parametrize("D,heads,Lq,Lk", [(64, 5, 256, 256), (64, 5, 200, 77), (64, 10, 1024, 1024), (40, 8, 256, 256),
                                           (80, 8, 256, 77), (160, 8, 64, 64), (160, 8, 256, 77), (40, 8, 4096, 4096), (64, 5, 2100, 2100), (64, 2, 4096, 77)])
def test_flash_attn(ops, gpu, D, heads, Lq, Lk):
    B = 2
    Cc = heads * D
    q, k, v = rnd(B, Lq, Cc, seed=1), rnd(B, Lk, Cc, seed=2), rnd(B, Lk, Cc, seed=3)
    # spike one key against one query so the running max jumps mid-sequence (exercises the rescale path)
    if Lk > 128:
        k[0, Lk - 70] = q[0, 5] * 4.0
    ref = _attn_ref(q, k, v, heads)
    Lkpad = (Lk + 63) // 64 * 64
    vt = torch.full((B, Cc, Lkpad), float("nan"), dtype=torch.float16)   # pad columns beyond roundup8(Lk) are never read
    vt[:, :, :(Lk + 7) // 8 * 8] = 0                                      # contract: [Lk, roundup8(Lk)) finite
    vt[:, :, :Lk] = v.half().permute(0, 2, 1)
    out = ops.flash_attn(q.half().reshape(B * Lq, Cc).to(gpu), Cc, k.half().reshape(B * Lk, Cc).to(gpu), Cc,
                         vt.to(gpu), Lkpad, B, heads, D, Lq, Lk)
    report("flash_attn D%d h%d Lq%d Lk%d" % (D, heads, Lq, Lk), rel_inf(out.reshape(B, Lq, Cc), ref))
    # folded form (ctrl_attn_desc.k_prescaled): K arrives multiplied by scale*log2(e) -- the product path's form; here the
    # pre-scaled K is rounded to fp16 a second time (the plan rounds once, in the projection's epilogue), hence the same bound
    ks = (k * (1.4426950408889634 / math.sqrt(D))).half()
    out2 = ops.flash_attn(q.half().reshape(B * Lq, Cc).to(gpu), Cc, ks.reshape(B * Lk, Cc).to(gpu), Cc,
                          vt.to(gpu), Lkpad, B, heads, D, Lq, Lk, k_prescaled=True)
    report("flash_attn folded D%d h%d Lq%d Lk%d" % (D, heads, Lq, Lk), rel_inf(out2.reshape(B, Lq, Cc), ref))


def test_flash_attn_d64_variants(ops, gpu):
    """csrc/attention_d64.hip: every instruction-selection variant of the head_dim-64 long-sequence kernel (and variant 0, the
    round-2 kernel) computes softmax(QK^T/sqrt(d))V -- a spiked key mid-sequence forces the rescale path, a spike in the very
    first tile the forced first update, Lq not a multiple of the workgroup's queries the clamped tail"""
    B, heads, D, L = 2, 3, 64, 2240          # 35 key tiles (not a multiple of the ring depth), 2240 = 8.75 x 256 queries
    Cc = heads * D
    q, k, v = rnd(B, L, Cc, seed=1), rnd(B, L, Cc, seed=2), rnd(B, L, Cc, seed=3)
    k[0, L - 70] = q[0, 5] * 4.0
    k[1, 3] = q[1, 100] * 6.0
    ref = _attn_ref(q, k, v, heads)
    vt = v.half().permute(0, 2, 1).contiguous()
    ks = (k * (1.4426950408889634 / math.sqrt(D))).half()
    try:
        for variant in range(0, 15):
            ops.set_attn_variant(variant)
            out = ops.flash_attn(q.half().reshape(B * L, Cc).to(gpu), Cc, ks.reshape(B * L, Cc).to(gpu), Cc, vt.to(gpu), L,
                                 B, heads, D, L, L, k_prescaled=True)
            report("flash_attn d64 variant %d" % variant, rel_inf(out.reshape(B, L, Cc), ref))
        with pytest.raises((ValueError, RuntimeError)):
            ops.set_attn_variant(99)
    finally:
        ops.set_attn_variant(-1)          # back to the default (CTRL_ATTN_VARIANT or the best measured)


def test_temporal_attn(ops, gpu):
    for Fr in (16, 14, 24):
        Bc, HW, heads = 2, 12, 5
        Cc = heads * 64
        qkv = rnd(Bc * Fr * HW, 3 * Cc, seed=Fr)
        q, k, v = qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:]

        def to_seq(t):  # [(b f p)][c] -> [(b p)][f][c]
            return t.reshape(Bc, Fr, HW, Cc).permute(0, 2, 1, 3).reshape(Bc * HW, Fr, Cc)
        ref = _attn_ref(to_seq(q), to_seq(k), to_seq(v), heads)
        ref = ref.reshape(Bc, HW, Fr, Cc).permute(0, 2, 1, 3).reshape(Bc * Fr * HW, Cc)
        out = ops.temporal_attn(qkv.half().to(gpu), Bc, Fr, HW, heads)
        report("temporal_attn F%d" % Fr, rel_inf(out, ref))


@pytest.mark.parametrize("Cc,hw,silu", [(320, 32 * 32, True), (640, 16 * 16, False), (1280, 64, True), (320, 128 * 128, True)])
def test_groupnorm(ops, gpu, Cc, hw, silu):
    n = 2
    x = rnd(n, hw, Cc, seed=1) + 0.5
    g, b = rnd(Cc, seed=2) + 1.0, rnd(Cc, seed=3)
    ref = F.group_norm(x.permute(0, 2, 1), 32, g, b, eps=1e-5).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    out = ops.groupnorm(x.half().to(gpu), g.to(gpu), b.to(gpu), n, hw, eps=1e-5, silu=silu)
    report("groupnorm C%d hw%d" % (Cc, hw), rel_inf(out, ref))
    out = ops.groupnorm(x.to(gpu), g.to(gpu), b.to(gpu), n, hw, eps=1e-5, silu=silu)          # fp32 stream input
    report("groupnorm C%d hw%d (fp32 in)" % (Cc, hw), rel_inf(out, ref))


@pytest.mark.parametrize("Cc", [512, 320, 640, 1280])
def test_layernorm(ops, gpu, Cc):
    x = rnd(1030, Cc, seed=1) * 2 + 0.3
    g, b = rnd(Cc, seed=2) + 1.0, rnd(Cc, seed=3)
    ref = F.layer_norm(x, (Cc,), g, b, eps=1e-5)
    out = ops.layernorm(x.half().to(gpu), g.to(gpu), b.to(gpu))
    report("layernorm C%d" % Cc, rel_inf(out, ref))
    out = ops.layernorm(x.to(gpu), g.to(gpu), b.to(gpu))                                         # fp32 stream input
    report("layernorm C%d (fp32 in)" % Cc, rel_inf(out, ref))


def test_layout_and_pool(ops, gpu):
    x = rnd(3, 70, 9, 11, seed=1)
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        y = ops.nchw_to_nhwc(x.to(dt).to(gpu))
        report("nchw_to_nhwc %s" % dt, rel_inf(y.permute(0, 3, 1, 2), x.to(dt).float()), 1e-3)
        z = ops.nhwc_to_nchw(x.permute(0, 2, 3, 1).contiguous().half().to(gpu), dtype=dt, scale=0.5)
        report("nhwc_to_nchw %s" % dt, rel_inf(z, x * 0.5), 5e-3)
    lat = rnd(4, 4, 128, 128, seed=2)
    report("avgpool 128->64", rel_inf(ops.avgpool_nchw(lat.to(gpu), 64, 64), F.adaptive_avg_pool2d(lat, (64, 64))), 1e-6)
    img = rnd(2, 3, 1024, 1024, seed=3)
    report("avgpool 1024->512", rel_inf(ops.avgpool_nchw(img.half().to(gpu), 512, 512), F.adaptive_avg_pool2d(img, (512, 512))), 1e-3)


def test_timestep_and_small_linear(ops, gpu):
    t = torch.tensor([999.0, 749.0, 1.0, 250.5])
    for dim in (320, 640, 1280):
        half = dim // 2
        freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
        arg = t[:, None] * freq[None]
        ref = torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)
        out = ops.timestep_sincos(t.to(gpu), 4, dim)
        err = (out.cpu() - ref).abs().max().item()
        print("PARITY sincos dim%d max_abs=%.3e" % (dim, err))
        assert err < 2e-4      # fp32 sin/cos of arguments up to 1e3
    x = rnd(16, 1280, seed=1)
    w, b = rnd(333, 1280, seed=2, scale=0.05), rnd(333, seed=3)
    wp = ops.pack_linear_w(w.to(gpu))
    out = ops.linear_small(x.to(gpu), wp, b.to(gpu), in_silu=True, out_silu=True)
    report("linear_small silu", rel_inf(out, F.silu(F.silu(x) @ w.t() + b)), 1e-5)
    out = ops.linear_small(x[:3].contiguous().to(gpu), wp, None)
    report("linear_small plain", rel_inf(out, x[:3] @ w.t()), 1e-5)


def test_blend_addvec(ops, gpu):
    a, b = rnd(100, 320, seed=1), rnd(100, 320, seed=2)
    mix = torch.tensor([0.3])
    al = torch.sigmoid(mix)
    report("blend", rel_inf(ops.blend(a.half().to(gpu), b.half().to(gpu), mix.to(gpu)), al * a + (1 - al) * b))
    report("blend fp32", rel_inf(ops.blend(a.to(gpu), b.to(gpu), mix.to(gpu)), al * a + (1 - al) * b), 1e-6)
    v = rnd(4, 320, seed=3)
    ref = a.reshape(4, 25, 320) + v[:, None]
    report("add_rowvec", rel_inf(ops.add_rowvec(a.half().to(gpu), v.to(gpu), 25, 4), ref.reshape(100, 320)))


@pytest.mark.parametrize("cin,cout,stride,nchw,h", [(3, 16, 1, True, 64), (4, 320, 1, True, 32), (16, 16, 1, False, 48),
                                                    (16, 32, 2, False, 48), (32, 32, 1, False, 32)])
def test_conv3x3_direct(ops, gpu, cin, cout, stride, nchw, h):
    n = 2
    x, w, b = rnd(n, cin, h, h, seed=1), rnd(cout, cin, 3, 3, seed=2, scale=0.1), rnd(cout, seed=3)
    ref = F.silu(F.conv2d(x, w, b, stride=stride, padding=1))
    wd = ops.pack_conv_w_direct(w.to(gpu))
    xin = x.to(gpu) if nchw else x.permute(0, 2, 3, 1).contiguous().half().to(gpu)
    out = ops.conv3x3_direct(xin, wd, b.to(gpu), cout, stride=stride, silu=True, nchw=nchw)
    report("conv3x3_direct %d->%d s%d" % (cin, cout, stride), rel_inf(out.permute(0, 3, 1, 2), ref))


def test_router(ops, gpu):
    wg = torch.randn(13, 3, generator=torch.Generator().manual_seed(1))
    for mask in ([1, 1, 1], [1, 0, 1]):
        lg = wg.clone()
        for e, m in enumerate(mask):
            if m == 0:
                lg[:, e] -= 1e6
        ref = torch.softmax(lg, dim=-1)
        out = ops.router_weights(wg.to(gpu), mask)
        report("router softmax mask=%s" % mask, rel_inf(out, ref), 1e-6)
    xs = [rnd(2, 320, 8, 8, seed=i) for i in range(3)]
    w = torch.softmax(wg, -1)
    ref = sum(w[4, e] * xs[e] for e in range(3))
    for dt in (torch.float32, torch.float16):
        out = ops.router_merge([x.to(dt).to(gpu) for x in xs], w[4].contiguous().to(gpu), [0, 1, 2])
        report("router merge %s" % dt, rel_inf(out, ref), 1e-6 if dt == torch.float32 else TOL)


@pytest.mark.parametrize("cin,cout,h,taps,n", [(320, 320, 16, 9, 2), (640, 1280, 8, 9, 2), (320, 320, 16, 1, 2), (1280, 1280, 8, 9, 8),
                                              (320, 320, 64, 9, 8)])
def test_split_operand_conv(ops, gpu, cin, cout, h, taps, n):
    """ctrl_igemm_desc.a_split: GroupNorm+SiLU writes [hi | lo] fp16 halves of its fp32 result and the convolution walks
    both -> the product is exact in A to ~2^-22 (fp16 operand rounding gone).  Both forms: 1 = weights packed twice, one
    long K axis; 2 = plain weights, (hi, lo) k-tile pairs sharing one staged weight tile (the default of the ControlNet
    plan), also through split-K.  Reference in fp64 on the un-rounded GroupNorm output; the plain fp16-operand path is
    printed beside it."""
    g = torch.Generator().manual_seed(7)
    x = torch.randn(n, h, h, cin, generator=g)                      # fp32 stream (not fp16-representable)
    gam, bet = 1 + 0.1 * torch.randn(cin, generator=g), 0.1 * torch.randn(cin, generator=g)
    k = 3 if taps == 9 else 1
    w = rnd(cout, cin, k, k, seed=2, scale=0.02)
    b = rnd(cout, seed=3)
    xn = F.silu(F.group_norm(x.permute(0, 3, 1, 2).double(), 32, gam.double(), bet.double(), eps=1e-5))
    ref = F.conv2d(xn, w.double(), b.double(), padding=k // 2).permute(0, 2, 3, 1)
    y2 = ops.groupnorm_split(x.to(gpu), gam.to(gpu), bet.to(gpu), n, h * h, eps=1e-5, silu=True)
    hi, lo = y2[..., :cin].float().cpu(), y2[..., cin:].float().cpu()
    report("gn split hi+lo vs fp64", rel_inf(hi.double() + lo.double(), xn.permute(0, 2, 3, 1)), 2e-6)
    geom = dict(Hin=h, Win=h, Hout=h, Wout=h, stride=1, up=1)
    M = n * h * h
    ws = torch.empty(16 * M * cout, dtype=torch.float32, device=gpu)
    for mode, wp in ((1, ops.pack_conv_w_dup(w.to(gpu))), (2, ops.pack_conv_w(w.to(gpu)))):
        for use_ws in (False, True):
            out = torch.empty(n, h, h, cout, dtype=torch.float32, device=gpu)
            mir = torch.empty(n, h, h, 2 * cout, dtype=torch.float16, device=gpu)
            ops.igemm(y2, 2 * cin, wp, M, cout, 2 * cin, taps=taps, mode=ops.IG_CONV2D, geom=geom,
                      bias=b.to(gpu), rows_per_img=h * h, segs=[(out, cout, 0, cout, ops.SEG_ROW, 1)], a_split=mode,
                      out16=mir, ld16=2 * cout, out16_lo_off=cout, splitk_ws=ws if use_ws else None)
            tag = "split-operand conv mode %d%s %dx%d taps%d M%d" % (mode, " +splitk scratch" if use_ws else "", cin, cout, taps, M)
            report(tag, rel_inf(out.double(), ref), 3e-5)
            report("   mirror hi+lo", rel_inf(mir[..., :cout].double().cpu() + mir[..., cout:].double().cpu(), ref), 3e-5)
    y1 = ops.groupnorm(x.to(gpu), gam.to(gpu), bet.to(gpu), n, h * h, eps=1e-5, silu=True)
    out1 = torch.empty(n, h, h, cout, dtype=torch.float32, device=gpu)
    ops.igemm(y1, cin, ops.pack_conv_w(w.to(gpu)), M, cout, cin, taps=taps, mode=ops.IG_CONV2D, geom=geom,
              bias=b.to(gpu), rows_per_img=h * h, segs=[(out1, cout, 0, cout, ops.SEG_ROW, 1)])
    print("PARITY   (plain fp16-operand conv beside it: rel_inf=%.3e)" % rel_inf(out1.double(), ref))


def test_frame_sharded_temporal_ops_are_bit_exact(ops, gpu):
    """the two frame-mixing kernels of a clip whose frames are sharded over ranks, given exact exchanges:
    temporal attention with LOCAL query frames against the all-gathered K|V rows, and the Conv3d over the halo-padded
    operand (ctrl_igemm_desc.t_pad) must reproduce the unsharded kernels bit for bit"""
    Bc, Fr, HW, heads, W = 2, 16, 20, 5, 4
    Cc, Fl = heads * 64, Fr // W
    qkv = rnd(Bc * Fr * HW, 3 * Cc, seed=5).half().to(gpu)
    full = ops.temporal_attn(qkv, Bc, Fr, HW, heads).reshape(Bc, Fr, HW, Cc)
    v5 = qkv.reshape(Bc, Fr, HW, 3 * Cc)
    kv_all = torch.stack([v5[:, r * Fl:(r + 1) * Fl, :, Cc:] for r in range(W)]).contiguous()      # [W][Bc][Fl][HW][2C]
    for r in range(W):
        q_loc = v5[:, r * Fl:(r + 1) * Fl, :, :Cc].reshape(Bc * Fl * HW, Cc).contiguous()
        out = ops.temporal_attn(q_loc, Bc, Fr, HW, heads, kv=kv_all.reshape(-1, 2 * Cc), Fq=Fl, Fl=Fl)
        assert torch.equal(out.reshape(Bc, Fl, HW, Cc), full[:, r * Fl:(r + 1) * Fl]), "rank %d" % r
    # Conv3d (3,1,1): padded operand [clip][Fl + 2][HW][C] with the neighbours' frames (zeros at the clip ends) in the halo slots
    Cc = 64
    x = rnd(Bc * Fr * HW, Cc, seed=6).half().to(gpu)
    w = rnd(Cc, Cc, 3, 1, 1, seed=2, scale=0.05)
    bias = rnd(Cc, seed=3).to(gpu)
    wp = ops.pack_conv_w(w.reshape(Cc, Cc, 3, 1).to(gpu))
    ref = torch.empty(Bc * Fr * HW, Cc, dtype=torch.float16, device=gpu)
    ops.igemm(x, Cc, wp, Bc * Fr * HW, Cc, Cc, taps=3, mode=ops.IG_TEMPORAL, bias=bias, segs=[(ref, Cc, 0, Cc, ops.SEG_ROW, 1)], F=Fr, HW=HW)
    ref = ref.reshape(Bc, Fr, HW, Cc)
    x5 = x.reshape(Bc, Fr, HW, Cc)
    for r in range(W):
        pad = torch.zeros(Bc, Fl + 2, HW, Cc, dtype=torch.float16, device=gpu)
        lo, hi = r * Fl - 1, (r + 1) * Fl + 1
        pad[:, (1 if lo < 0 else 0):(Fl + 1 if hi > Fr else Fl + 2)] = x5[:, max(lo, 0):min(hi, Fr)]
        out = torch.empty(Bc * Fl * HW, Cc, dtype=torch.float16, device=gpu)
        ops.igemm(pad, Cc, wp, Bc * Fl * HW, Cc, Cc, taps=3, mode=ops.IG_TEMPORAL, bias=bias, segs=[(out, Cc, 0, Cc, ops.SEG_ROW, 1)],
                  F=Fl, HW=HW, t_pad=True)
        assert torch.equal(out.reshape(Bc, Fl, HW, Cc), ref[:, r * Fl:(r + 1) * Fl]), "rank %d" % r
    print("PARITY frame-sharded temporal attention / Conv3d: bit-exact vs unsharded (4 shards)")


@pytest.mark.parametrize("K,geglu,f32", [(320, False, True), (2048, False, True), (512, True, False)])
def test_two_workgroup_tiles_at_large_m(ops, gpu, K, geglu, f32):
    """the 128x256 / 256x128 tiles (two workgroups per CU) the dispatcher picks for epilogue-heavy token GEMMs at
    M >= 65536: fp32 stream update (fp32 residual in, fp32 master + fp16 mirror out) and GEGLU"""
    M, N = 65536 + 40, (1024 if geglu else 512)
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3)
    wp, bp = ops.pack_linear_w(w.to(gpu), geglu=geglu), ops.pack_vec(b.to(gpu), geglu=geglu)
    y = x @ w.t() + b
    if geglu:
        ref = y[:, :N // 2] * F.gelu(y[:, N // 2:])
        out = ops.linear(x.half().to(gpu), wp, bias=bp, geglu=True)
        report("2-WG tile geglu K%d" % K, rel_inf(out, ref))
    else:
        r = torch.randn(M, N, generator=torch.Generator().manual_seed(4))
        ref = y + r
        out = torch.empty(M, N, dtype=torch.float32, device=gpu)
        mirror = torch.empty(M, N, dtype=torch.float16, device=gpu)
        ops.igemm(x.half().to(gpu), K, wp, M, N, K, bias=bp, res=r.to(gpu), ldres=N, segs=[(out, N, 0, N, ops.SEG_ROW, 1)], out16=mirror, ld16=N)
        report("2-WG tile f32 stream K%d master" % K, rel_inf(out, ref), 2e-5)
        report("2-WG tile f32 stream K%d mirror" % K, rel_inf(mirror, ref))


def test_tile_walk_orders_are_bit_identical(ops, gpu):
    """csrc/tile_order.h only decides WHICH workgroup computes a tile: every order must give the same bits (GEGLU with bias;
    fp32 stream update with fp32 residual + fp16 mirror; a ragged M), and the fp32 result matches the fp32 reference"""
    try:
        for (M, N, K, geglu) in [(16384, 4096, 512, True), (16384 - 24, 2048, 320, False)]:
            x, w, b = rnd(M, K, seed=11), rnd(N, K, seed=12, scale=0.05), rnd(N, seed=13)
            wp, bp = ops.pack_linear_w(w.to(gpu), geglu=geglu), ops.pack_vec(b.to(gpu), geglu=geglu)
            xg = x.half().to(gpu)
            r = torch.randn(M, N, generator=torch.Generator().manual_seed(14)).to(gpu)

            def run():
                if geglu:
                    return (ops.linear(xg, wp, bias=bp, geglu=True),)
                out = torch.empty(M, N, dtype=torch.float32, device=gpu)
                mirror = torch.empty(M, N, dtype=torch.float16, device=gpu)
                ops.igemm(xg, K, wp, M, N, K, bias=bp, res=r, ldres=N, segs=[(out, N, 0, N, ops.SEG_ROW, 1)], out16=mirror, ld16=N)
                return out, mirror
            ops.set_igemm_order("legacy")
            base = run()
            for spec in ["auto", "m,1", "m,3", "n,0", "n,2", "m,4"]:
                ops.set_igemm_order(spec)
                got = run()
                for g, want in zip(got, base):
                    assert torch.equal(g, want), (M, N, K, spec)
            if not geglu:
                ref = x.half().float() @ w.half().float().t() + b + r.cpu()
                report("tile walk orders, fp32 stream", rel_inf(base[0], ref), 2e-5)
        with pytest.raises(ValueError):
            ops.set_igemm_order("sideways")
    finally:
        ops.set_igemm_order("auto")


# ---------------------------------------------------------------------------------------------------------------------
# The 8-phase wide-tile kernel (csrc/igemm.hip: igemm8_kernel).  The dispatcher only picks it where the grid fills the chip,
# i.e. for none of the small shapes above: ctrl_igemm_set_wide(2) routes every eligible problem through it.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture
def wide(ops):
    ops.set_igemm_wide(2)
    yield
    ops.set_igemm_wide(-1)


@pytest.mark.parametrize("name", ["test_linear", "test_conv3x3", "test_conv3x3_splitk", "test_split_operand_conv", "test_linear_geglu",
                                  "test_linear_f32_stream", "test_conv1x1_nchw_out_scaled", "test_conv3x3_with_half_resolution_residual",
                                  "test_single_transposed_output_epilogue", "test_qkv_segments", "test_conv1x1_upsampled_shortcut"])
def test_wide_tile_kernel_runs_the_op_tests(ops, gpu, wide, name):
    """every parametrisation of the GEMM / convolution op tests above, through igemm8_kernel (rows, conv2d with stride / folded
    up-sampling / half-resolution residual, split-K, both split-operand walks, GEGLU, fp32 streams, transposed outputs, segments)"""
    fn = globals()[name]
    marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
    if not marks:
        fn(ops, gpu)
        return
    assert len(marks) == 1
    names = [n.strip() for n in marks[0].args[0].split(",")]
    for vals in marks[0].args[1]:
        vals = vals if isinstance(vals, (tuple, list)) else (vals,)
        fn(ops, gpu, **dict(zip(names, vals)))


@pytest.mark.parametrize("Cc,Fr,HW,b", [(320, 5, 48, 2), (640, 16, 16, 2), (256, 3, 100, 1)])
def test_wide_tile_temporal_conv(ops, gpu, wide, Cc, Fr, HW, b):
    """Conv3d (3,1,1) over frames at the adapters' channel counts (the op test above has C = 64, below the wide tile): NI = 5 (320, 640)
    and NI = 4 (256) tiles, clips whose first / last frames have no neighbour, ragged M"""
    x = rnd(b, Cc, Fr, HW, 1, seed=1)
    w = rnd(Cc, Cc, 3, 1, 1, seed=2, scale=0.03)
    bias = rnd(Cc, seed=3)
    ref = F.conv3d(x, w, bias, padding=(1, 0, 0))
    xr = x[..., 0].permute(0, 2, 3, 1).contiguous().reshape(b * Fr * HW, Cc)
    wp = ops.pack_conv_w(w.reshape(Cc, Cc, 3, 1).to(gpu))
    out = torch.empty(b * Fr * HW, Cc, dtype=torch.float16, device=gpu)
    ops.igemm(xr.half().to(gpu), Cc, wp, b * Fr * HW, Cc, Cc, taps=3, mode=ops.IG_TEMPORAL, bias=bias.to(gpu),
              segs=[(out, Cc, 0, Cc, ops.SEG_ROW, 1)], F=Fr, HW=HW)
    refr = ref[..., 0].permute(0, 2, 3, 1).reshape(b * Fr * HW, Cc)
    report("wide-tile temporal conv3 C%d F%d" % (Cc, Fr), rel_inf(out, refr))


def test_wide_tile_frame_sharded_conv3d_matches_unsharded(ops, gpu, wide):
    """the halo-padded operand of a frame-sharded clip (ctrl_igemm_desc.t_pad) through the wide tile: bit-identical to the
    unsharded convolution of the same kernel"""
    Bc, Fr, HW, W, Cc = 2, 16, 24, 4, 320
    Fl = Fr // W
    x = rnd(Bc * Fr * HW, Cc, seed=6).half().to(gpu)
    w = rnd(Cc, Cc, 3, 1, 1, seed=2, scale=0.03)
    bias = rnd(Cc, seed=3).to(gpu)
    wp = ops.pack_conv_w(w.reshape(Cc, Cc, 3, 1).to(gpu))
    ref = torch.empty(Bc * Fr * HW, Cc, dtype=torch.float16, device=gpu)
    ops.igemm(x, Cc, wp, Bc * Fr * HW, Cc, Cc, taps=3, mode=ops.IG_TEMPORAL, bias=bias, segs=[(ref, Cc, 0, Cc, ops.SEG_ROW, 1)], F=Fr, HW=HW)
    ref = ref.reshape(Bc, Fr, HW, Cc)
    x5 = x.reshape(Bc, Fr, HW, Cc)
    for r in range(W):
        pad = torch.zeros(Bc, Fl + 2, HW, Cc, dtype=torch.float16, device=gpu)
        lo, hi = r * Fl - 1, (r + 1) * Fl + 1
        pad[:, (1 if lo < 0 else 0):(Fl + 1 if hi > Fr else Fl + 2)] = x5[:, max(lo, 0):min(hi, Fr)]
        out = torch.empty(Bc * Fl * HW, Cc, dtype=torch.float16, device=gpu)
        ops.igemm(pad, Cc, wp, Bc * Fl * HW, Cc, Cc, taps=3, mode=ops.IG_TEMPORAL, bias=bias, segs=[(out, Cc, 0, Cc, ops.SEG_ROW, 1)],
                  F=Fl, HW=HW, t_pad=True)
        assert torch.equal(out.reshape(Bc, Fl, HW, Cc), ref[:, r * Fl:(r + 1) * Fl]), "rank %d" % r


def test_wide_tile_is_bit_reproducible_and_close_to_the_ring_kernels(ops, gpu):
    """same problem through both kernel families: each bit-identical run to run; between them the fp32 summation order differs
    (64- vs 32-deep k-tiles), so the fp16 outputs agree to one rounding"""
    M, N, K = 2048, 640, 1280
    x = rnd(M, K, seed=3).half().to(gpu)
    w = rnd(N, K, seed=4, scale=0.03)
    wp = ops.pack_linear_w(w.to(gpu))
    outs = []
    for mode in (0, 2, 2):
        ops.set_igemm_wide(mode)
        outs.append(ops.linear(x, wp))
    ops.set_igemm_wide(-1)
    assert torch.equal(outs[1], outs[2])
    report("wide tile vs ring kernel", rel_inf(outs[1], outs[0].float().cpu()), 1e-3)


@pytest.mark.parametrize("cin,cout,h,n,res_up", [(640, 640, 32, 8, 0), (320, 320, 64, 8, 0), (1280, 1280, 32, 8, 0), (640, 640, 32, 8, 2),
                                                 (1280, 1280, 8, 8, 0), (1280, 1280, 16, 8, 0), (640, 640, 16, 8, 0), (1280, 640, 16, 3, 0)])
def test_splitk_reduced_inside_the_launch(ops, gpu, cin, cout, h, n, res_up):
    """The K-splits of the wide tile are summed by the last-arriving workgroup of every tile (agent-scope release / ticket / acquire,
    csrc/igemm.hip) instead of a finish kernel: against the fp32 reference, and bit-identical from run to run (the slabs are added
    in a fixed order whoever arrives last) -- 4, 2, 2 and 4 splits in the first four cases (one level), then 15, 8, 8 and 16 splits
    (round 6: two levels, groups of four; the ControlNet's 8^2 / 16^2 convolutions at b = 8), with bias, time vector and a
    (half-resolution) residual.  No splitk_finish launch may be left."""
    x = rnd(n, cin, h, h, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2, scale=0.02)
    b, temb = rnd(cout, seed=3), rnd(n, cout, seed=4)
    hr = h // 2 if res_up == 2 else h
    r = torch.randn(n, hr, hr, cout, generator=torch.Generator().manual_seed(5))
    r_full = r.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2) if res_up == 2 else r
    ref = F.conv2d(x, w, b, padding=1) + temb[:, :, None, None] + r_full.permute(0, 3, 1, 2)
    wp = ops.pack_conv_w(w.to(gpu))
    xh = x.permute(0, 2, 3, 1).contiguous().half().to(gpu)
    ws = torch.empty(20 * max(n * h * h, 256) * cout, dtype=torch.float32, device=gpu)      # (tile-padded slabs + ticket words)
    prev = ops.set_policy("CTRL_SPLITK_INLAUNCH", "all")      # (the two-level form is opt-in: measured slower than the finish kernel)
    try:
        _splitk_inlaunch_body(ops, gpu, xh, wp, cout, b, temb, r, res_up, ws, ref, cin, h, n)
    finally:
        ops.set_policy("CTRL_SPLITK_INLAUNCH", prev)


def _splitk_inlaunch_body(ops, gpu, xh, wp, cout, b, temb, r, res_up, ws, ref, cin, h, n):
    outs = []
    for _ in range(4):
        ws.fill_(float("nan"))                       # a slab read before it was written would poison the result
        outs.append(ops.conv2d(xh, wp, cout, taps=9, bias=b.to(gpu), rowvec=temb.to(gpu), res=r.to(gpu), res_up=res_up, splitk_ws=ws))
    with ops.Profiler() as prof:
        ops.conv2d(xh, wp, cout, taps=9, bias=b.to(gpu), rowvec=temb.to(gpu), res=r.to(gpu), res_up=res_up, splitk_ws=ws)
    assert any("in-launch" in rec[2] for rec in prof.launches) and "splitk_finish" not in prof.rows, (prof.rows.keys(), [rec[2] for rec in prof.launches])
    report("in-launch split-K conv %d->%d @%d n%d" % (cin, cout, h, n), rel_inf(outs[0].permute(0, 3, 1, 2), ref))
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "split-K result differs from run to run"


def test_range_check_flags_an_activation_beyond_fp16(ops, gpu):
    """CTRL_CHECK_FINITE=1 / ctrl_range_check(1): a GEMM whose result leaves the fp16 range (|x| > 65504) raises the per-device flag from
    its epilogue -- fp16 rows, the fp16 mirror of an fp32 stream and transposed fp16 outputs -- and an in-range one does not; with the
    check off nothing is flagged.  Then the module path: a ControlNet forward on weights scaled out of range raises RuntimeError."""
    from ctrl_adapter_amd import _lib as L
    lib = L.lib()
    M, N, K = 512, 320, 512
    x = torch.full((M, K), 16.0).half().to(gpu)
    w_small = ops.pack_linear_w(torch.full((N, K), 0.01).to(gpu))          # 16 * 0.01 * 512 = 81.9: fine
    w_big = ops.pack_linear_w(torch.full((N, K), 16.0).to(gpu))            # 16 * 16 * 512 = 131072 > 65504
    assert lib.ctrl_range_check(0) == 0 and lib.ctrl_range_status(1) == 2  # off: status says so
    ops.linear(x, w_big)
    assert lib.ctrl_range_check(1) == 1
    try:
        assert lib.ctrl_range_status(1) == 0                               # nothing was recorded while it was off
        ops.linear(x, w_small)
        assert lib.ctrl_range_status(1) == 0
        y = ops.linear(x, w_big)
        assert torch.isinf(y).any()
        assert lib.ctrl_range_status(1) == 1 and lib.ctrl_range_status(1) == 0      # raised, then reset
        for wide in (2, 0):                                                # both kernel families share the epilogue
            ops.set_igemm_wide(wide)
            ops.linear(x, w_big)
            assert lib.ctrl_range_status(1) == 1, wide
            out = torch.empty(2, N, 256, dtype=torch.float16, device=gpu)   # transposed fp16 output ([img][C][tokens])
            ops.igemm(x, K, w_big, M, N, K, segs=[(out, 256, 0, N, ops.SEG_TRANSPOSED, 256)])
            assert lib.ctrl_range_status(1) == 1, wide
            o32 = torch.empty(M, N, dtype=torch.float32, device=gpu)       # fp32 stream + fp16 mirror: the mirror overflows
            mir = torch.empty(M, N, dtype=torch.float16, device=gpu)
            ops.igemm(x, K, w_big, M, N, K, segs=[(o32, N, 0, N, ops.SEG_ROW, 1)], out16=mir, ld16=N)
            assert lib.ctrl_range_status(1) == 1 and torch.isfinite(o32).all() and torch.isinf(mir).any(), wide
        ops.set_igemm_wide(-1)
        # module level: RuntimeError from the mirror's forward
        import helpers  # noqa: F401  (puts tests/golden on the path)
        import ctrl_adapter_amd as P
        import cases
        from oracle.init import seeded_init
        cn = seeded_init(P.ControlNetModel(**cases.CONTROLNET_KW), seed=11)
        with torch.no_grad():
            for p_ in cn.parameters():
                if p_.ndim == 4 and p_.shape[1] >= 320:
                    p_.mul_(3000.0)                                         # conv weights far out of the synthetic range
        cn = cn.to(gpu)
        inp = cases.controlnet_inputs(N=1, hs=8)
        with pytest.raises(RuntimeError, match="non-finite fp16 activation"):
            cn(inp["sample"].half().to(gpu), inp["timestep"].to(gpu), inp["encoder_hidden_states"].half().to(gpu),
               inp["controlnet_cond"].half().to(gpu), return_dict=False)
    finally:
        ops.set_igemm_wide(-1)
        lib.ctrl_range_check(0)


@pytest.mark.parametrize("B,Ltok,K,Cc,wide", [(4, 1024, 512, 320, 2), (4, 1024, 512, 320, 1), (2, 256, 512, 1280, 1), (8, 64, 1280, 1280, 1),
                                              (2, 4096, 320, 320, 2), (3, 136, 512, 640, 1)])
def test_qkv_one_launch_is_bit_identical_to_two(ops, gpu, B, Ltok, K, Cc, wide):
    """Round 5: Q | K | V^T of a self-attention projection in ONE launch on the vector epilogue (row-major Q|K segment + transposed V
    segment: ctrl_igemm_desc::seg) -- the same values bit for bit as the Q|K launch + the V launch of rounds 1-4, K pre-scaled through
    the bounded scale2 range (scale2_from / scale2_to).  wide = 2 forces the 8-phase tile, 1 leaves the dispatcher alone (ring tiles
    for the small grids: 128- and 64-wide tiles divide 2 * Cc)."""
    M = B * Ltok
    x = rnd(M, K, seed=1).half().to(gpu)
    w = rnd(3 * Cc, K, seed=2, scale=0.05)
    wp = ops.pack_linear_w(w.to(gpu))
    Lpad = (Ltok + 63) // 64 * 64
    ks = 1.4426950408889634 / math.sqrt(64.0)
    ops.set_igemm_wide(wide)
    try:
        qk1 = torch.zeros(M, 2 * Cc, dtype=torch.float16, device=gpu)
        vt1 = torch.zeros(B, Cc, Lpad, dtype=torch.float16, device=gpu)
        ops.igemm(x, K, wp, M, 3 * Cc, K, scale2=ks, scale2_from=Cc, scale2_to=2 * Cc,
                  segs=[(qk1, 2 * Cc, 0, 2 * Cc, ops.SEG_ROW, 1), (vt1, Lpad, 2 * Cc, Cc, ops.SEG_TRANSPOSED, Ltok)])
        qk2 = torch.zeros_like(qk1)
        vt2 = torch.zeros_like(vt1)
        ops.igemm(x, K, wp, M, 2 * Cc, K, scale2=ks, scale2_from=Cc, segs=[(qk2, 2 * Cc, 0, 2 * Cc, ops.SEG_ROW, 1)])
        ops.igemm(x, K, wp[2 * Cc:], M, Cc, K, segs=[(vt2, Lpad, 0, Cc, ops.SEG_TRANSPOSED, Ltok)])
    finally:
        ops.set_igemm_wide(-1)
    ref = x.float().cpu() @ w.t()
    ref[:, Cc:2 * Cc] *= ks
    report("qkv one launch q|k (B%d L%d C%d)" % (B, Ltok, Cc), rel_inf(qk1, ref[:, :2 * Cc]))
    report("qkv one launch v^T", rel_inf(vt1[:, :, :Ltok], ref[:, 2 * Cc:].reshape(B, Ltok, Cc).permute(0, 2, 1)))
    assert torch.equal(qk1, qk2) and torch.equal(vt1, vt2)
    if Lpad > Ltok:
        assert vt1[:, :, Ltok:].abs().max().item() == 0.0


@pytest.mark.parametrize("Cc,hw,silu,n", [(320, 32 * 32, True, 3), (640, 16 * 16, False, 2), (1280, 64, True, 8), (1280, 7 * 9, False, 2),
                                          (640, 32 * 32, True, 2), (320, 100, False, 1)])
def test_groupnorm_fused_small_maps(ops, gpu, Cc, hw, silu, n):
    """Round 5: statistics + apply of a small GroupNorm(32) map in ONE launch (gn_fused_kernel: a workgroup owns 80 channels = 8 / 4 / 2
    whole groups of one image).  Against torch's group_norm, fp16 and fp32 inputs, plain and split [hi | lo] results; the large map
    of test_groupnorm does not qualify and stays on the two-kernel form."""
    x = (rnd(n, hw, Cc, seed=1) * 1.7 + 0.5).half().float()          # fp16-representable: both input dtypes see the same values
    g, b = rnd(Cc, seed=2) + 1.0, rnd(Cc, seed=3)
    ref = F.group_norm(x.permute(0, 2, 1), 32, g, b, eps=1e-6).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    for xin in (x.half().to(gpu), x.to(gpu)):
        out = ops.groupnorm_fused(xin, g.to(gpu), b.to(gpu), n, hw, eps=1e-6, silu=silu)
        assert out is not None
        report("groupnorm fused C%d hw%d (%s in)" % (Cc, hw, str(xin.dtype)[6:]), rel_inf(out, ref))
        sp = ops.groupnorm_fused(xin, g.to(gpu), b.to(gpu), n, hw, eps=1e-6, silu=silu, split=True)
        assert torch.equal(sp[..., :Cc], out)
        report("groupnorm fused split hi+lo", rel_inf(sp[..., :Cc].float() + sp[..., Cc:].float(), ref), 2e-5 if xin.dtype == torch.float32 else 2e-5)
    assert ops.groupnorm_fused(torch.zeros(1, 128 * 128, 320, dtype=torch.float32, device=gpu), g[:320].to(gpu), b[:320].to(gpu), 1, 128 * 128) is None


@pytest.mark.parametrize("cin,cout,stride,h,w_", [(16, 16, 1, 48, 48), (16, 32, 2, 48, 48), (32, 32, 1, 32, 32), (16, 16, 1, 37, 53), (32, 16, 2, 35, 41),
                                                  (16, 32, 1, 16, 520)])
def test_conv3x3_small_mfma(ops, gpu, cin, cout, stride, h, w_):
    """Round 5: the conditioning embedder's 16 / 32-channel 3x3 convolutions on the matrix cores (smallconv.hip:
    conv3x3_small_mfma_kernel) against torch's conv2d + SiLU; ragged widths / heights exercise the masked tail lanes and rows."""
    n = 3
    x, w, b = rnd(n, cin, h, w_, seed=1), rnd(cout, cin, 3, 3, seed=2, scale=0.1), rnd(cout, seed=3)
    ref = F.silu(F.conv2d(x, w, b, stride=stride, padding=1))
    wp = ops.pack_conv_w(w.to(gpu))
    xin = x.permute(0, 2, 3, 1).contiguous().half().to(gpu)
    out = ops.conv3x3_small_mfma(xin, wp, b.to(gpu), cout, stride=stride, silu=True)
    report("conv3x3_small_mfma %d->%d s%d %dx%d" % (cin, cout, stride, h, w_), rel_inf(out.permute(0, 3, 1, 2), ref))
    wd = ops.pack_conv_w_direct(w.to(gpu))
    old = ops.conv3x3_direct(xin, wd, b.to(gpu), cout, stride=stride, silu=True, nchw=False)
    assert rel_inf(out, old) < 1e-3


# ---------------------------------------------------------------------------------------------------------------------------
# Round 6: the fused GEGLU feed-forward (csrc/ffn.hip)
# ---------------------------------------------------------------------------------------------------------------------------
def _ffn_case(M, seed, gpu, ops, scale_w=0.03):
    x = rnd(M, 512, seed=seed)
    w1, b1 = rnd(4096, 512, seed=seed + 1, scale=scale_w), rnd(4096, seed=seed + 2, scale=0.1)
    w2, b2 = rnd(512, 2048, seed=seed + 3, scale=scale_w), rnd(512, seed=seed + 4, scale=0.1)
    r = torch.randn(M, 512, generator=torch.Generator().manual_seed(seed + 5))          # fp32 stream: NOT fp16-representable
    w1p, b1p = ops.pack_linear_w(w1.to(gpu), geglu=True), ops.pack_vec(b1.to(gpu), geglu=True)
    w2p = ops.pack_linear_w(w2.to(gpu))
    return x, w1, b1, w2, b2, r, w1p, b1p, w2p


def _ffn_ref(x, w1, b1, w2, b2, r, chunk=16384):
    """fp32 torch reference of FeedForward(GEGLU) + residual (diffusers GEGLU: hidden * gelu(gate), exact erf), on the fp16-rounded
    operands, in row chunks (M = 131072 x 4096 fp32 would be 2 GB at once)"""
    xh, w1h, w2h = x.half().float(), w1.half().float(), w2.half().float()
    out = torch.empty(x.shape[0], 512)
    for i in range(0, x.shape[0], chunk):
        y = xh[i:i + chunk] @ w1h.t() + b1
        h = y[:, :2048] * F.gelu(y[:, 2048:])
        out[i:i + chunk] = h @ w2h.t() + b2 + r[i:i + chunk]
    return out


@pytest.mark.parametrize("M", [128, 1000, 32768, 131072])
def test_ffn_fused_vs_fp32_torch_and_two_launch_form(ops, gpu, M):
    """ONE launch for LayerNorm'd tokens -> GEGLU projection -> output projection + bias + fp32 residual (csrc/ffn.hip), against fp32
    torch (<= 3e-4 rel-inf: the hidden activation is rounded to fp16 once, exactly as the two-launch form rounds it) and against the
    two-launch form of igemm.hip on the same packed weights; M = 1000: a ragged last tile; fp32 master + fp16 mirror out."""
    torch.set_num_threads(8)
    x, w1, b1, w2, b2, r, w1p, b1p, w2p = _ffn_case(M, 7000 + M % 97, gpu, ops)
    w2pp = ops.ffn_pack_w2(w2p)
    xg, rg, b2g = x.half().to(gpu), r.to(gpu), b2.to(gpu)
    out = torch.full((M, 512), float("nan"), dtype=torch.float32, device=gpu)
    mir = torch.full((M, 512), float("nan"), dtype=torch.float16, device=gpu)
    ops.ffn_fused(xg, w1p, b1p, w2pp, b2=b2g, res=rg, out=out, out16=mir)
    # the two-launch form
    hid = ops.linear(xg, w1p, bias=b1p, geglu=True)
    out2 = torch.empty(M, 512, dtype=torch.float32, device=gpu)
    ops.igemm(hid, 2048, w2p, M, 512, 2048, bias=b2g, res=rg, ldres=512, segs=[(out2, 512, 0, 512, ops.SEG_ROW, 1)])
    torch.cuda.synchronize()
    e2 = rel_inf(out, out2.cpu())
    ref = _ffn_ref(x, w1, b1, w2, b2, r)
    report("ffn fused M%d vs fp32 torch" % M, rel_inf(out, ref), 3e-4)
    report("ffn fused M%d fp16 mirror" % M, rel_inf(mir, ref))
    report("ffn fused M%d vs two launches" % M, e2, 1e-4)
    # deterministic: a second launch gives the same bits
    outb = torch.empty_like(out)
    ops.ffn_fused(xg, w1p, b1p, w2pp, b2=b2g, res=rg, out=outb)
    assert torch.equal(out, outb)


def test_ffn_fused_fp16_stream_and_no_residual(ops, gpu):
    """the other epilogue forms the plans use: fp16 token stream in and out (the SDXL adapter's default), and no residual / bias"""
    M = 4096 + 64
    x, w1, b1, w2, b2, r, w1p, b1p, w2p = _ffn_case(M, 7100, gpu, ops)
    w2pp = ops.ffn_pack_w2(w2p)
    xg = x.half().to(gpu)
    r16 = r.half()
    o16 = ops.ffn_fused(xg, w1p, b1p, w2pp, b2=b2.to(gpu), res=r16.to(gpu))
    report("ffn fused fp16 stream", rel_inf(o16, _ffn_ref(x, w1, b1, w2, b2, r16.float())))
    o0 = ops.ffn_fused(xg, w1p, b1p, w2pp)
    report("ffn fused plain", rel_inf(o0, _ffn_ref(x, w1, b1, w2, torch.zeros(512), torch.zeros(M, 512))))


@pytest.mark.parametrize("M", [16384, 131072])
def test_ffn_fused_under_wave_jitter_is_bit_identical(ops, gpu, M):
    """The fused kernel's intra-workgroup protocol (counted vmcnt waits, re-staging one phase after the last read, the P hand-over between
    the wave columns) under UNEVEN wave timing: the jitter build of the kernel (CTRL_FF_FUSED=jitter) makes every wave sleep a pseudo-random
    time at the head of every phase segment.  Its result must equal the plain kernel's bit for bit, on inputs where neighbouring workgroups
    hold different rows (a stale LDS slot then holds OTHER data, not a lucky copy), several rounds."""
    x, w1, b1, w2, b2, r, w1p, b1p, w2p = _ffn_case(M, 7300, gpu, ops)
    w2pp = ops.ffn_pack_w2(w2p)
    xg, rg, b2g = x.half().to(gpu), r.to(gpu), b2.to(gpu)
    ref = torch.empty(M, 512, dtype=torch.float32, device=gpu)
    ops.ffn_fused(xg, w1p, b1p, w2pp, b2=b2g, res=rg, out=ref)
    prev = ops.set_policy("CTRL_FF_FUSED", "jitter")
    try:
        for rnd_ in range(3):
            out = torch.full((M, 512), float("nan"), dtype=torch.float32, device=gpu)
            ops.ffn_fused(xg, w1p, b1p, w2pp, b2=b2g, res=rg, out=out)
            torch.cuda.synchronize()
            nbad = int((out != ref).sum().item())
            assert nbad == 0, "round %d: %d of %d values differ under wave jitter (worst %.3e)" % (rnd_, nbad, out.numel(), (out - ref).abs().max().item())
    finally:
        ops.set_policy("CTRL_FF_FUSED", prev)
    print("PARITY ffn fused M%d under wave jitter: 3 rounds bit-identical" % M)
