"""Thin Python wrappers over the op-level C ABI (ctrl_op_*): allocate outputs with torch, pass raw pointers.

Used by the parity tests and micro-benchmarks; the product forward path goes through the plan-level
entry points (controlnet.py / ctrl_adapter.py) and never composes these from Python.
"""
import ctypes as C

import torch

from . import _lib as L

IG_ROWS, IG_CONV2D, IG_TEMPORAL = 0, 1, 2
SEG_ROW, SEG_TRANSPOSED = 0, 1


def _f16(*shape, device="cuda"):
    return torch.empty(*shape, dtype=torch.float16, device=device)


def pack_conv_w(w):
    """[Cout][Cin][kh][kw] -> fp16 [Cout][kh*kw][Cin]"""
    cout, cin = w.shape[0], w.shape[1]
    taps = w[0, 0].numel()
    out = _f16(cout, taps * cin)
    L.check(L.lib().ctrl_op_pack_conv_w(L.ptr(w.contiguous()), L.dtype_code(w.dtype), L.ptr(out), cout, cin, taps, L.cur_stream()))
    return out


def pack_conv_w_dup(w):
    """[Cout][Cin][kh][kw] -> fp16 [Cout][kh*kw][2*Cin] (split-operand convolution: every tap's weights twice)"""
    cout, cin = w.shape[0], w.shape[1]
    taps = w[0, 0].numel()
    out = _f16(cout, taps * 2 * cin)
    L.check(L.lib().ctrl_op_pack_conv_w_dup(L.ptr(w.contiguous()), L.dtype_code(w.dtype), L.ptr(out), cout, cin, taps, L.cur_stream()))
    return out


def pack_conv_w_direct(w):
    cout, cin = w.shape[0], w.shape[1]
    out = torch.empty(9, cin, cout, dtype=torch.float32, device=w.device)
    L.check(L.lib().ctrl_op_pack_conv_w_direct(L.ptr(w.contiguous()), L.dtype_code(w.dtype), L.ptr(out), cout, cin, L.cur_stream()))
    return out


def pack_linear_w(w, geglu=False):
    n, k = w.shape
    out = _f16(n, k)
    L.check(L.lib().ctrl_op_pack_linear_w(L.ptr(w.contiguous()), L.dtype_code(w.dtype), L.ptr(out), n, k, int(geglu), L.cur_stream()))
    return out


def pack_vec(v, geglu=False):
    out = torch.empty(v.numel(), dtype=torch.float32, device=v.device)
    L.check(L.lib().ctrl_op_pack_vec(L.ptr(v.contiguous()), L.dtype_code(v.dtype), L.ptr(out), v.numel(), int(geglu), L.cur_stream()))
    return out


def igemm(A, lda, W, M, Nout, Cin, taps=1, mode=IG_ROWS, geom=None, bias=None, rowvec=None, rows_per_img=1,
          res=None, ldres=0, scale=1.0, geglu=False, segs=None, F=0, HW=0, out16=None, ld16=0, splitk_ws=None,
          blend_mix=None, blend_x=None, ld_blend=0, a_split=False, out16_lo_off=0, t_pad=False, scale2=0.0, scale2_from=0,
          res_up=0, scale2_to=0):
    """segs: list of (out_tensor, ld, col_begin, ncols, fmt, L); res_up=2: `res` is [N][Hout/2][Wout/2][ldres], read through a
    nearest x2 up-sampling (conv2d)"""
    d = L.IGemmDesc()
    d.A = A.data_ptr(); d.lda = lda; d.mode = mode; d.Cin = Cin; d.taps = taps
    g = geom or {}
    d.Hin = g.get("Hin", 1); d.Win = g.get("Win", 1); d.Hout = g.get("Hout", 1); d.Wout = g.get("Wout", 1)
    d.stride = g.get("stride", 1); d.up = g.get("up", 1)
    d.F = F; d.HW = HW; d.t_pad = int(t_pad)
    d.W = W.data_ptr(); d.M = M; d.Nout = Nout; d.Ktot = taps * Cin
    d.bias = bias.data_ptr() if bias is not None else None
    d.rowvec = rowvec.data_ptr() if rowvec is not None else None
    d.rowvec_ld = rowvec.shape[-1] if rowvec is not None else 0
    d.rows_per_img = rows_per_img
    d.res = res.data_ptr() if res is not None else None
    d.ldres = ldres
    d.res_f32 = int(res is not None and res.dtype == torch.float32)
    d.res_up = res_up
    d.out16 = out16.data_ptr() if out16 is not None else None
    d.splitk_ws = splitk_ws.data_ptr() if splitk_ws is not None else None
    d.splitk_ws_bytes = splitk_ws.numel() * splitk_ws.element_size() if splitk_ws is not None else 0
    d.ld16 = ld16
    d.blend_mix = blend_mix.data_ptr() if blend_mix is not None else None
    d.blend_x = blend_x.data_ptr() if blend_x is not None else None
    d.ld_blend = ld_blend
    d.blend_f32 = int(blend_x is not None and blend_x.dtype == torch.float32)
    d.scale = scale; d.geglu = int(geglu)
    d.scale2 = scale2; d.scale2_from = scale2_from; d.scale2_to = scale2_to
    d.a_split = int(a_split); d.out16_lo_off = out16_lo_off      # a_split: 0 | 1 (weights packed twice) | 2 (paired walk)
    d.nseg = len(segs)
    for i, (out, ld, cb, nc, fmt, Ltok) in enumerate(segs):
        d.seg[i].out = out.data_ptr(); d.seg[i].ld = ld; d.seg[i].col_begin = cb; d.seg[i].ncols = nc
        d.seg[i].fmt = fmt; d.seg[i].dtype = L.dtype_code(out.dtype); d.seg[i].L = Ltok
    L.check(L.lib().ctrl_op_igemm(C.byref(d), L.cur_stream()))


def ffn_pack_w2(w2_packed):
    """fp16 [512][2048] linear pack of ff.net.2 -> the layout / k order of the fused feed-forward kernel (csrc/ffn.hip)"""
    N, K = w2_packed.shape
    out = torch.empty_like(w2_packed)
    L.check(L.lib().ctrl_op_ffn_pack_w2(L.ptr(w2_packed), L.ptr(out), N, K, L.cur_stream()))
    return out


def ffn_fused(x, w1p, b1p, w2pp, b2=None, res=None, out=None, out16=None):
    """out = res + ((x W1h^T + b1h) * gelu(x W1g^T + b1g)) W2^T + b2 in ONE launch (csrc/ffn.hip): x fp16 [M][512]; w1p / b1p the GEGLU
    packs of ff.net.0.proj (pack_linear_w / pack_vec with geglu=True), w2pp = ffn_pack_w2(pack_linear_w(W2)); res fp32 or fp16 [M][512];
    out fp32 or fp16 [M][512] (default fp16), out16 an optional fp16 mirror of an fp32 out"""
    M, K = x.shape
    if out is None:
        out = _f16(M, 512)
    d = L.FfnDesc()
    d.X = x.data_ptr(); d.ldx = K; d.W1 = w1p.data_ptr(); d.b1 = b1p.data_ptr(); d.W2p = w2pp.data_ptr()
    o = d.out
    o.M = M; o.Nout = 512; o.Ktot = 2048; o.Cin = 2048; o.taps = 1; o.scale = 1.0; o.rows_per_img = 1
    o.Hin = o.Win = o.Hout = o.Wout = o.stride = o.up = 1
    o.bias = b2.data_ptr() if b2 is not None else None
    o.res = res.data_ptr() if res is not None else None
    o.ldres = 512
    o.res_f32 = int(res is not None and res.dtype == torch.float32)
    o.out16 = out16.data_ptr() if out16 is not None else None
    o.ld16 = 512
    o.nseg = 1
    o.seg[0].out = out.data_ptr(); o.seg[0].ld = 512; o.seg[0].col_begin = 0; o.seg[0].ncols = 512
    o.seg[0].fmt = SEG_ROW; o.seg[0].dtype = L.dtype_code(out.dtype); o.seg[0].L = 1
    L.check(L.lib().ctrl_op_ffn(C.byref(d), L.cur_stream()))
    return out


def set_igemm_order(spec):
    """tile walk of the implicit GEMM over the XCDs: "auto" (default) | "legacy" | "m,G" | "n,G" (csrc/tile_order.h); results
    do not depend on it"""
    if L.lib().ctrl_igemm_set_order(spec.encode()) != 0:
        raise ValueError("unknown tile walk order %r" % (spec,))


def set_igemm_wide(mode):
    """which GEMM / convolution problems run on the 8-phase wide-tile kernel: 0 none, 1 where the grid fills the chip (default), 2 every
    eligible problem (tests), -1 back to the default (csrc/igemm.hip)"""
    L.check(L.lib().ctrl_igemm_set_wide(int(mode)))


def set_group_launches(mode):
    """grouped launches over the sibling adapter blocks of a pyramid level (csrc/ops.h: OpCollector): 1 / True on (default), 2 on with
    the tile selection every problem would get alone (bit-identical to the one-by-one forward), 0 / False off, None queries; returns
    the mode."""
    return int(L.lib().ctrl_group_launches(-1 if mode is None else int(mode)))


def set_policy(name, value):
    """overrides one CTRL_* variable of the library's policy table for this process (csrc/policy.h; value None = unset, as if the
    variable were absent from the environment); returns the previous value (None = unset).  Test / experiment hook."""
    prev = L.lib().ctrl_policy_get(name.encode())
    L.check(L.lib().ctrl_policy_set(name.encode(), None if value is None else str(value).encode()))
    return None if prev is None else prev.decode()


def policy():
    """{name: value} of every CTRL_* variable that is set (environment snapshot + overrides)"""
    lib = L.lib()
    out = {}
    for i in range(lib.ctrl_policy_count()):
        n = lib.ctrl_policy_name(i)
        v = lib.ctrl_policy_get(n)
        if v is not None:
            out[n.decode()] = v.decode()
    return out


def set_attn_variant(v):
    """instruction-selection variant of the head_dim-64 long-sequence attention kernel (0 = the round-2 kernel); every
    variant computes the same function (csrc/attention_d64.hip)"""
    L.check(L.lib().ctrl_attn_set_variant(int(v)))


def linear(x, w_packed, bias=None, res=None, geglu=False):
    """x [M][K] fp16 -> [M][N] fp16 (N/2 for GEGLU)"""
    M, K = x.shape
    N = w_packed.shape[0]
    on = N // 2 if geglu else N
    out = _f16(M, on)
    igemm(x, K, w_packed, M, N, K, bias=bias, res=res, ldres=on, geglu=geglu,
          segs=[(out, on, 0, on, SEG_ROW, 1)])
    return out


def conv2d(x_nhwc, w_packed, Cout, taps=9, stride=1, up=1, bias=None, rowvec=None, res=None, scale=1.0,
           out_nchw_dtype=None, splitk_ws=None, res_up=0):
    """x [N][H][W][Cin] fp16 -> [N][Ho][Wo][Cout] fp16 (or NCHW in out_nchw_dtype)"""
    N, H, W, Cin = x_nhwc.shape
    pad = 1 if taps == 9 else 0
    Ho = (H * up + 2 * pad - (3 if taps == 9 else 1)) // stride + 1
    Wo = (W * up + 2 * pad - (3 if taps == 9 else 1)) // stride + 1
    M = N * Ho * Wo
    geom = dict(Hin=H, Win=W, Hout=Ho, Wout=Wo, stride=stride, up=up)
    if out_nchw_dtype is None:
        out = _f16(N, Ho, Wo, Cout)
        segs = [(out, Cout, 0, Cout, SEG_ROW, 1)]
    else:
        out = torch.empty(N, Cout, Ho, Wo, dtype=out_nchw_dtype, device=x_nhwc.device)
        segs = [(out, Ho * Wo, 0, Cout, SEG_TRANSPOSED, Ho * Wo)]
    igemm(x_nhwc, Cin, w_packed, M, Cout, Cin, taps=taps, mode=IG_CONV2D, geom=geom, bias=bias, rowvec=rowvec,
          rows_per_img=Ho * Wo, res=res, ldres=Cout, scale=scale, segs=segs, splitk_ws=splitk_ws, res_up=res_up)
    return out


def flash_attn(Q, ldq, K, ldk, Vt, Lkpad, B, heads, D, Lq, Lk, scale=None, kvB=0, k_prescaled=False):
    out = _f16(B * Lq, heads * D)
    d = L.AttnDesc()
    d.Q = Q.data_ptr(); d.ldq = ldq; d.K = K.data_ptr(); d.ldk = ldk; d.Vt = Vt.data_ptr(); d.Lkpad = Lkpad
    d.O = out.data_ptr(); d.ldo = heads * D
    d.B = B; d.heads = heads; d.D = D; d.Lq = Lq; d.Lk = Lk; d.kvB = kvB if kvB else B
    d.scale = scale if scale is not None else D ** -0.5
    d.k_prescaled = int(k_prescaled)
    L.check(L.lib().ctrl_op_flash_attn(C.byref(d), L.cur_stream()))
    return out


def temporal_attn(qkv, Bc, F, HW, heads, kv=None, Fq=0, Fl=0):
    """qkv [(b f) p][3C] (q | k | v) -> [(b f) p][C]; sharded form: qkv holds the LOCAL Fq query frames as [..][C] rows
    and kv = the gathered K|V rows [world][Bc][Fl][HW][2C]"""
    Cc = heads * 64
    out = _f16(qkv.shape[0], Cc)
    d = L.TAttnDesc()
    d.Q = qkv.data_ptr(); d.ld = qkv.shape[1]; d.O = out.data_ptr(); d.ldo = Cc
    d.Bc = Bc; d.F = F; d.HW = HW; d.heads = heads; d.scale = 64 ** -0.5
    d.Fq = Fq; d.Fl = Fl
    if kv is not None:
        d.KV = kv.data_ptr(); d.ldkv = kv.shape[-1]
    L.check(L.lib().ctrl_op_temporal_attn(C.byref(d), L.cur_stream()))
    return out


def groupnorm(x, gamma, beta, imgs, rows_per_img, G=32, eps=1e-5, silu=False):
    Cc = x.shape[-1]
    lib = L.lib()
    stats = torch.zeros(lib.ctrl_op_gn_stats_floats(imgs, rows_per_img, Cc, G), dtype=torch.float32, device=x.device)
    y = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    L.check(lib.ctrl_op_gn_stats(L.ptr(x), L.dtype_code(x.dtype), L.ptr(stats), imgs, rows_per_img, Cc, G, L.cur_stream()))
    L.check(lib.ctrl_op_gn_apply(L.ptr(x), L.dtype_code(x.dtype), L.ptr(stats), L.ptr(gamma), L.ptr(beta), L.ptr(y), imgs, rows_per_img, Cc, G,
                                 C.c_float(eps), int(silu), L.cur_stream()))
    return y


def groupnorm_fused(x, gamma, beta, imgs, rows_per_img, G=32, eps=1e-5, silu=False, split=False):
    """GroupNorm(32) of a small map in one launch (csrc/norm.hip: gn_fused_kernel); None when the problem does not qualify"""
    Cc = x.shape[-1]
    lib = L.lib()
    if not lib.ctrl_op_gn_fused_applies(L.dtype_code(x.dtype), rows_per_img, Cc, G):
        return None
    y = torch.empty(tuple(x.shape[:-1]) + ((2 * Cc) if split else Cc,), dtype=torch.float16, device=x.device)
    L.check(lib.ctrl_op_gn_fused(L.ptr(x), L.dtype_code(x.dtype), L.ptr(gamma), L.ptr(beta), L.ptr(y), C.c_int64(2 * Cc if split else Cc),
                                 Cc if split else 0, imgs, rows_per_img, Cc, G, C.c_float(eps), int(silu), L.cur_stream()))
    return y


def groupnorm_split(x, gamma, beta, imgs, rows_per_img, G=32, eps=1e-5, silu=False):
    """GroupNorm whose result is a split operand: [rows][2C] = [hi | lo], hi + lo == the fp32 result to ~2^-22"""
    Cc = x.shape[-1]
    lib = L.lib()
    stats = torch.zeros(lib.ctrl_op_gn_stats_floats(imgs, rows_per_img, Cc, G), dtype=torch.float32, device=x.device)
    y = torch.empty(tuple(x.shape[:-1]) + (2 * Cc,), dtype=torch.float16, device=x.device)
    L.check(lib.ctrl_op_gn_stats(L.ptr(x), L.dtype_code(x.dtype), L.ptr(stats), imgs, rows_per_img, Cc, G, L.cur_stream()))
    L.check(lib.ctrl_op_gn_apply_split(L.ptr(x), L.dtype_code(x.dtype), L.ptr(stats), L.ptr(gamma), L.ptr(beta), L.ptr(y),
                                       C.c_int64(2 * Cc), Cc, imgs, rows_per_img, Cc, G, C.c_float(eps), int(silu), L.cur_stream()))
    return y


def layernorm(x, gamma, beta, eps=1e-5):
    M, Cc = x.shape
    y = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    L.check(L.lib().ctrl_op_layernorm(L.ptr(x), L.dtype_code(x.dtype), C.c_int64(Cc), L.ptr(gamma), L.ptr(beta), L.ptr(y), C.c_int64(Cc), M, Cc,
                                      C.c_float(eps), L.cur_stream()))
    return y


def nchw_to_nhwc(x):
    N, Cc, H, W = x.shape
    y = _f16(N, H, W, Cc)
    L.check(L.lib().ctrl_op_nchw_to_nhwc(L.ptr(x.contiguous()), L.dtype_code(x.dtype), L.ptr(y), N, Cc, H * W, L.cur_stream()))
    return y


def nhwc_to_nchw(x, dtype=torch.float32, scale=1.0):
    N, H, W, Cc = x.shape
    y = torch.empty(N, Cc, H, W, dtype=dtype, device=x.device)
    L.check(L.lib().ctrl_op_nhwc_to_nchw(L.ptr(x), L.ptr(y), L.dtype_code(dtype), N, Cc, H * W, C.c_float(scale), L.cur_stream()))
    return y


def avgpool_nchw(x, Hout, Wout):
    """exact integer-ratio F.adaptive_avg_pool2d; dtype preserved"""
    N, Cc, H, W = x.shape
    y = torch.empty(N, Cc, Hout, Wout, dtype=x.dtype, device=x.device)
    L.check(L.lib().ctrl_avgpool_nchw(L.ptr(x.contiguous()), L.ptr(y), L.dtype_code(x.dtype), N * Cc, H, W, Hout, Wout, L.cur_stream()))
    return y


def timestep_sincos(t, N, dim):
    out = torch.empty(N, dim, dtype=torch.float32, device=t.device)
    L.check(L.lib().ctrl_op_timestep_sincos(L.ptr(t), t.numel(), L.ptr(out), N, dim, L.cur_stream()))
    return out


def linear_small(x, w_packed, bias=None, in_silu=False, out_silu=False):
    M, K = x.shape
    N = w_packed.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    L.check(L.lib().ctrl_op_linear_small(L.ptr(x), C.c_int64(K), L.ptr(w_packed), L.ptr(bias), L.ptr(out), C.c_int64(N), M, N, K,
                                         int(in_silu), int(out_silu), L.cur_stream()))
    return out


def blend(xs, xt, mix):
    y = torch.empty_like(xs)
    L.check(L.lib().ctrl_op_blend(L.ptr(xs), L.dtype_code(xs.dtype), L.ptr(xt), L.dtype_code(xt.dtype), L.ptr(mix), L.ptr(y),
                                  L.dtype_code(y.dtype), C.c_size_t(xs.numel()), L.cur_stream()))
    return y


def add_rowvec(x, v, rows_per_img, vmod):
    M, Cc = x.shape
    y = torch.empty_like(x)
    L.check(L.lib().ctrl_op_add_rowvec(L.ptr(x), L.dtype_code(x.dtype), L.ptr(v), C.c_int64(v.shape[-1]), L.ptr(y), L.dtype_code(y.dtype),
                                       C.c_size_t(M), Cc, rows_per_img, vmod, L.cur_stream()))
    return y


def conv3x3_direct(x, w_direct, bias, Cout, stride=1, silu=False, nchw=False):
    if nchw:
        N, Cin, H, W = x.shape
    else:
        N, H, W, Cin = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = _f16(N, Ho, Wo, Cout)
    L.check(L.lib().ctrl_op_conv3x3_direct(L.ptr(x), L.dtype_code(x.dtype), int(nchw), L.ptr(w_direct), L.ptr(bias), L.ptr(out),
                                           N, Cin, Cout, H, W, stride, int(silu), L.cur_stream()))
    return out


def conv3x3_small_mfma(x, w_packed, bias, Cout, stride=1, silu=False):
    """x NHWC fp16 [N][H][W][Cin] (Cin, Cout in {16, 32}), w_packed = pack_conv_w(w): the MFMA form of the small convolutions"""
    N, H, W, Cin = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = _f16(N, Ho, Wo, Cout)
    L.check(L.lib().ctrl_op_conv3x3_small_mfma(L.ptr(x), L.ptr(w_packed), L.ptr(bias), L.ptr(out), N, Cin, Cout, H, W, stride, int(silu),
                                               L.cur_stream()))
    return out


def router_weights(wg, mask, equal_weights=False):
    R, E = wg.shape
    out = torch.empty(R, E, dtype=torch.float32, device=wg.device)
    m = (C.c_int * E)(*[int(v) for v in mask]) if mask is not None else None
    L.check(L.lib().ctrl_router_weights(L.ptr(wg), m, L.ptr(out), R, E, int(equal_weights), L.cur_stream()))
    return out


def router_merge(experts, weights_row, widx):
    K = len(experts)
    out = torch.empty_like(experts[0])
    ptrs = (C.c_void_p * K)(*[e.data_ptr() for e in experts])
    idx = (C.c_int * K)(*widx)
    L.check(L.lib().ctrl_router_merge(ptrs, L.ptr(weights_row), idx, K, L.ptr(out), L.dtype_code(out.dtype),
                                      C.c_size_t(out.numel()), L.cur_stream()))
    return out


_profiling = False


def profiling():
    """True while a Profiler block records (one kernel at a time: the mirrors keep to one stream then)"""
    return _profiling


class Profiler:
    """HIP-event profiler over kernel classes (see ctrl_prof_* in include/ctrl_hip.h)."""

    def __enter__(self):
        global _profiling
        L.check(L.lib().ctrl_prof_begin())
        _profiling = True
        return self

    def __exit__(self, *exc):
        global _profiling
        _profiling = False
        lib = L.lib()
        L.check(lib.ctrl_prof_end())
        self.rows = {}
        name = C.create_string_buffer(64)
        ms, fl, by = C.c_double(), C.c_double(), C.c_double()
        n = C.c_int()
        for i in range(lib.ctrl_prof_count()):
            L.check(lib.ctrl_prof_get(i, name, 64, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
            self.rows[name.value.decode()] = (ms.value, n.value, fl.value, by.value)
        # per-launch records, in launch order: (class tag, kernel symbol, shape note, ms, flops, bytes, grid work-items)
        self.launches = []
        tag, sym, det = C.create_string_buffer(64), C.create_string_buffer(256), C.create_string_buffer(256)
        grid = C.c_int64()
        for i in range(lib.ctrl_prof_launch_count()):
            L.check(lib.ctrl_prof_launch_get(i, tag, 64, sym, 256, det, 256, C.byref(ms), C.byref(fl), C.byref(by), C.byref(grid)))
            self.launches.append((tag.value.decode(), sym.value.decode(), det.value.decode(), ms.value, fl.value, by.value, grid.value))
        return False

    def per_kernel(self, reps=1):
        """launches grouped by (kernel symbol, shape): {key: dict(tag, symbol, shape, launches_per_step, ms_per_step,
        avg_launch_ms, flops_per_launch, bytes_per_launch, tflops, gbs, grid)} -- one row per template instantiation x shape"""
        acc = {}
        for tag, sym, det, ms, fl, by, grid in self.launches:
            k = (sym, det)
            a = acc.setdefault(k, dict(tag=tag, symbol=sym, shape=det, n=0, ms=0.0, flops=0.0, bytes=0.0, grid=grid))
            a["n"] += 1; a["ms"] += ms; a["flops"] += fl; a["bytes"] += by
        out = {}
        for (sym, det), a in acc.items():
            n, ms = a["n"], a["ms"]
            out[(sym + " " + det).strip()] = dict(
                tag=a["tag"], symbol=sym, shape=det, launches_per_step=n / reps, ms_per_step=ms / reps, avg_launch_ms=ms / n,
                flops_per_launch=a["flops"] / n, bytes_per_launch=a["bytes"] / n, grid=a["grid"],
                tflops=(a["flops"] / (ms * 1e-3) / 1e12) if a["flops"] and ms > 0 else None,
                gbs=(a["bytes"] / (ms * 1e-3) / 1e9) if a["bytes"] and ms > 0 else None)
        return out
