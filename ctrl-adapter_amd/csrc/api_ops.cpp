// extern "C" op-level entry points of libctrlhip (thin: validate, cast, forward to the op layer).
#include "ops.h"

#define S(stream) ((hipStream_t)(stream))
#define H(p) ((half_t*)(p))
#define CH(p) ((const half_t*)(p))

extern "C" {
int ctrl_op_igemm(const ctrl_igemm_desc* d, void* stream) {
    CTRL_CHECK(d != nullptr, "igemm: null descriptor");
    return op_igemm(*d, S(stream));
}
int ctrl_op_ffn(const ctrl_ffn_desc* d, void* stream) {
    CTRL_CHECK(d != nullptr, "ffn: null descriptor");
    return op_ffn_fused_group(d, 1, S(stream));
}
int ctrl_op_ffn_pack_w2(const void* w2_packed, void* out, int N, int K, void* stream) {
    CTRL_CHECK(w2_packed && out, "ffn_pack_w2: null argument");
    return op_ffn_pack_w2((const half_t*)w2_packed, (half_t*)out, N, K, S(stream));
}
int ctrl_igemm_set_order(const char* spec) { return igemm_set_order(spec); }
int ctrl_igemm_set_wide(int mode) {
    CTRL_CHECK(mode >= -1 && mode <= 2, "igemm_set_wide: mode must be -1 (default), 0 (never), 1 (auto) or 2 (every eligible problem)");
    return igemm_set_wide(mode);
}
int ctrl_igemm_tile_of(int bid, int ntm, int ntn, int mode, int group, int* tile_m, int* tile_n) {
    CTRL_CHECK(tile_m && tile_n && ntm > 0 && ntn > 0 && bid >= 0 && bid < ntm * ntn, "igemm_tile_of: bad arguments");
    igemm_tile_of(bid, ntm, ntn, mode, group, tile_m, tile_n);
    return 0;
}
int ctrl_op_flash_attn(const ctrl_attn_desc* d, void* stream) {
    CTRL_CHECK(d != nullptr, "flash_attn: null descriptor");
    return op_flash_attn(*d, S(stream));
}
int ctrl_attn_set_variant(int v) {
    CTRL_CHECK(v >= -1 && v <= 14, "attn_set_variant: variant out of range (-1 = default, 0 = round-2 kernel, 1..14 = attention_d64.hip)");
    return attn_set_variant(v);
}
int ctrl_attn_work_map(int gbid, int qtiles, int pairs, int* pair, int* qtile) {
    if (!pair || !qtile || gbid < 0 || qtiles < 1 || pairs < 1) return 0;
    return attn_work_map(gbid, qtiles, pairs, pair, qtile) ? 1 : 0;
}
int ctrl_op_temporal_attn(const ctrl_tattn_desc* d, void* stream) {
    CTRL_CHECK(d != nullptr, "temporal_attn: null descriptor");
    return op_temporal_attn(*d, S(stream));
}
size_t ctrl_op_gn_stats_floats(int imgs, int rows_per_img, int C, int G) { return op_gn_stats_floats(imgs, rows_per_img, C, G); }
int ctrl_op_gn_stats(const void* x, int x_dtype, float* stats, int imgs, int rows_per_img, int C, int G, void* stream) {
    return op_gn_stats(x, x_dtype, stats, imgs, rows_per_img, C, G, S(stream));
}
int ctrl_op_gn_apply(const void* x, int x_dtype, const float* stats, const float* gamma, const float* beta, void* y,
                     int imgs, int rows_per_img, int C, int G, float eps, int silu, void* stream) {
    return op_gn_apply(x, x_dtype, stats, gamma, beta, H(y), imgs, rows_per_img, C, G, eps, silu, S(stream));
}
int ctrl_op_gn_fused_applies(int x_dtype, int rows_per_img, int C, int G) { return op_gn_fused_fits(x_dtype, rows_per_img, C, G) ? 1 : 0; }
int ctrl_op_gn_fused(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int64_t ldy, int lo_off,
                     int imgs, int rows_per_img, int C, int G, float eps, int silu, void* stream) {
    return op_gn_fused(x, x_dtype, gamma, beta, H(y), imgs, rows_per_img, C, G, eps, silu, S(stream), (long)ldy, lo_off);
}
int ctrl_op_gn_apply_split(const void* x, int x_dtype, const float* stats, const float* gamma, const float* beta, void* y,
                           int64_t ldy, int lo_off, int imgs, int rows_per_img, int C, int G, float eps, int silu, void* stream) {
    return op_gn_apply(x, x_dtype, stats, gamma, beta, H(y), imgs, rows_per_img, C, G, eps, silu, S(stream), (long)ldy, lo_off);
}
int ctrl_op_layernorm(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy,
                      int M, int C, float eps, void* stream) {
    return op_layernorm(x, x_dtype, ldx, gamma, beta, H(y), ldy, M, C, eps, S(stream));
}
int ctrl_op_nchw_to_nhwc(const void* x, int dtype, void* y, int N, int C, int HW, void* stream) {
    return op_nchw_to_nhwc(x, dtype, H(y), N, C, HW, S(stream));
}
int ctrl_op_nhwc_to_nchw(const void* x, void* y, int dtype, int N, int C, int HW, float scale, void* stream) {
    return op_nhwc_to_nchw(CH(x), y, dtype, N, C, HW, scale, S(stream));
}
int ctrl_avgpool_nchw(const void* x, void* y, int dtype, int NC, int Hin, int Win, int Hout, int Wout, void* stream) {
    return op_avgpool_nchw(x, y, dtype, NC, Hin, Win, Hout, Wout, S(stream));
}
int ctrl_op_timestep_sincos(const float* t, int t_count, float* out, int N, int dim, void* stream) {
    return op_timestep_sincos(t, t_count, out, N, dim, S(stream));
}
int ctrl_op_linear_small(const float* x, int64_t ldx, const void* w, const float* b, float* out, int64_t ldo,
                         int M, int N, int K, int in_silu, int out_silu, void* stream) {
    return op_linear_small(x, ldx, CH(w), b, out, ldo, M, N, K, in_silu, out_silu, S(stream));
}
int ctrl_op_blend(const void* xs, int xs_dtype, const void* xt, int xt_dtype, const float* mix, void* y, int y_dtype, size_t n, void* stream) {
    return op_blend(xs, xs_dtype, xt, xt_dtype, mix, y, y_dtype, n, S(stream));
}
int ctrl_op_add_rowvec(const void* x, int x_dtype, const float* v, int64_t ldv, void* y, int y_dtype, size_t M, int C, int rows_per_img, int vmod, void* stream) {
    return op_add_rowvec(x, x_dtype, v, ldv, y, y_dtype, M, C, rows_per_img, vmod, S(stream));
}
int ctrl_op_conv3x3_direct(const void* in, int in_dtype, int in_nchw, const float* w, const float* bias, void* out,
                           int N, int Cin, int Cout, int Hin, int Win, int stride, int silu, void* stream) {
    return op_conv3x3_direct(in, in_dtype, in_nchw, w, bias, H(out), N, Cin, Cout, Hin, Win, stride, silu, S(stream));
}
int ctrl_op_conv3x3_small_mfma(const void* x, const void* w, const float* bias, void* out, int N, int Cin, int Cout, int Hin, int Win,
                               int stride, int silu, void* stream) {
    return op_conv3x3_small_mfma((const half_t*)x, (const half_t*)w, bias, H(out), N, Cin, Cout, Hin, Win, stride, silu, S(stream));
}
int ctrl_op_pack_conv_w(const void* w, int dtype, void* out, int Cout, int Cin, int taps, void* stream) {
    return op_pack_conv_w(w, dtype, H(out), Cout, Cin, taps, S(stream));
}
int ctrl_op_pack_conv_w_dup(const void* w, int dtype, void* out, int Cout, int Cin, int taps, void* stream) {
    return op_pack_conv_w_dup(w, dtype, H(out), Cout, Cin, taps, S(stream));
}
int ctrl_op_pack_conv_w_direct(const void* w, int dtype, float* out, int Cout, int Cin, void* stream) {
    return op_pack_conv_w_direct(w, dtype, out, Cout, Cin, S(stream));
}
int ctrl_op_pack_linear_w(const void* w, int dtype, void* out, int N, int K, int geglu, void* stream) {
    return op_pack_linear_w(w, dtype, H(out), N, K, geglu, S(stream));
}
int ctrl_op_pack_vec(const void* v, int dtype, float* out, int N, int geglu, void* stream) {
    return op_pack_vec(v, dtype, out, N, geglu, S(stream));
}
int ctrl_router_weights(const float* wg, const int* mask_host, float* weights_out, int R, int E, int equal_weights, void* stream) {
    return op_router_softmax(wg, mask_host, weights_out, R, E, equal_weights, S(stream));
}
int ctrl_router_merge(const void* const* experts_host, const float* weights_row, const int* widx_host, int K,
                      void* out, int dtype, size_t n, void* stream) {
    return op_weighted_merge(experts_host, weights_row, widx_host, K, out, dtype, n, S(stream));
}
int ctrl_prepare_images(const void* src_u8, int F, int Hin, int Win, const int32_t* hbounds, const int32_t* hk, int hks,
                        const int32_t* vbounds, const int32_t* vk, int vks, void* tmp_u8, void* out, int out_dtype,
                        int W, int H, int repeat, int cfg, void* stream) {
    return op_prepare_images((const unsigned char*)src_u8, F, Hin, Win, hbounds, hk, hks, vbounds, vk, vks, (unsigned char*)tmp_u8,
                             out, out_dtype, W, H, repeat, cfg, S(stream));
}
}
