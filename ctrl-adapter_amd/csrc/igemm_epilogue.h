// Epilogue of an accumulated implicit-GEMM tile, shared by every main loop of igemm.hip and by the fused feed-forward kernel (ffn.hip):
// bias, per-image vector, GEGLU, SiLU, residual, scale, AlphaBlender blend, fp32 master + fp16 mirror, row-major / transposed segments.
// Device code only; include inside an anonymous namespace of a .hip file after ops.h.
#pragma once

// Residual row of output row `row`.  res_up == 2: the residual is held at HALF the output resolution and read through a
// nearest x2 up-sampling -- the 1x1 shortcut convolution of the adapter's ResnetBlock2D commutes with the nearest up-sampling
// in front of it (model/resnet_block_2d.py:174-184 up-samples input_tensor, :216 applies conv_shortcut: every output pixel of
// the conv is the conv of ONE input pixel), so the shortcut runs on the quarter-size map and this epilogue fetches
// res[n][oy/2][ox/2]: the same values bit for bit, a quarter of the shortcut's FLOPs and of its fp32 round trip.
__device__ __forceinline__ size_t res_row_of(const IGemmArgs& e, int row) {
    if (e.res_up != 2) return (size_t)row;
    const int hw = e.Hout * e.Wout;
    const int n = row / hw, rem = row - n * hw;
    const int oy = rem / e.Wout, ox = rem - oy * e.Wout;
    return ((size_t)n * (e.Hout >> 1) + (oy >> 1)) * (size_t)(e.Wout >> 1) + (ox >> 1);
}

// range check (ctrl_igemm_desc::nonfinite): raise the flag when a value about to be rounded to fp16 does not fit (inf / nan / |x| > 65504)
__device__ __forceinline__ void flag_nonfinite(int32_t* flag, bool bad) {
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) *(volatile int32_t*)flag = 1;
}
__device__ __forceinline__ bool out_of_half(float x) { return !(fabsf(x) <= 65504.0f); }

// Eight residual values of output row `row` from column `col` on, as fp32 (the fp32 stream or an fp16 tensor; `up2`: held at half
// the output resolution and read through a nearest x2 up-sampling, see res_row_of).  32-bit element offsets (checked by op_igemm).
struct ResSrc { const void* p; unsigned ld; bool f32, up2; int hw, w, hh, wh; };
__device__ __forceinline__ void res_fetch8(const ResSrc& r, int row, int col, f4& a0, f4& a1) {
    unsigned rrow = (unsigned)row;
    if (r.up2) {
        const int n = row / r.hw, rem = row - n * r.hw;
        const int oy = rem / r.w, ox = rem - oy * r.w;
        rrow = (unsigned)((n * r.hh + (oy >> 1)) * r.wh + (ox >> 1));
    }
    const unsigned o = rrow * r.ld + (unsigned)col;
    if (r.f32) {
        const float* rp = (const float*)r.p + o;
        a0 = *(const f4*)rp;
        a1 = *(const f4*)(rp + 4);
    } else {
        const h8 rr = *(const h8*)((const half_t*)r.p + o);
        a0 = f4{(float)rr[0], (float)rr[1], (float)rr[2], (float)rr[3]};
        a1 = f4{(float)rr[4], (float)rr[5], (float)rr[6], (float)rr[7]};
    }
}

// scale of output column `col`: scale2 inside [scale2_from, scale2_to) (scale2_to == 0: to the end), scale elsewhere
__device__ __forceinline__ float col_scale(const IGemmArgs& e, int col) {
    return (e.scale2_from > 0 && col >= e.scale2_from && (e.scale2_to == 0 || col < e.scale2_to)) ? e.scale2 : e.scale;
}

// Segment `si` of the descriptor with every field selected VALUE by value from scalars that are opaque to the optimiser: left to
// itself the compiler turns "select between loaded kernel-argument fields" into "load from a selected address", which moves the whole
// descriptor to scratch for the entire kernel (every argument read then waits on vmcnt)
__device__ __forceinline__ IGemmSeg seg_select(const IGemmArgs& e, int si) {
    void* q_out[3] = {e.seg[0].out, e.seg[1].out, e.seg[2].out};
    int64_t q_ld[3] = {e.seg[0].ld, e.seg[1].ld, e.seg[2].ld};
    int q_cb[3] = {e.seg[0].col_begin, e.seg[1].col_begin, e.seg[2].col_begin}, q_nc[3] = {e.seg[0].ncols, e.seg[1].ncols, e.seg[2].ncols};
    int q_fmt[3] = {e.seg[0].fmt, e.seg[1].fmt, e.seg[2].fmt}, q_dt[3] = {e.seg[0].dtype, e.seg[1].dtype, e.seg[2].dtype}, q_L[3] = {e.seg[0].L, e.seg[1].L, e.seg[2].L};
    const int32_t* q_map[3] = {e.seg[0].img_map, e.seg[1].img_map, e.seg[2].img_map};
#pragma unroll
    for (int k = 0; k < 3; ++k) asm volatile("" : "+s"(q_out[k]), "+s"(q_ld[k]), "+s"(q_cb[k]), "+s"(q_nc[k]), "+s"(q_fmt[k]), "+s"(q_dt[k]), "+s"(q_L[k]), "+s"(q_map[k]));
    IGemmSeg sg;
    sg.out = si == 0 ? q_out[0] : (si == 1 ? q_out[1] : q_out[2]);
    sg.ld = si == 0 ? q_ld[0] : (si == 1 ? q_ld[1] : q_ld[2]);
    sg.col_begin = si == 0 ? q_cb[0] : (si == 1 ? q_cb[1] : q_cb[2]);
    sg.ncols = si == 0 ? q_nc[0] : (si == 1 ? q_nc[1] : q_nc[2]);
    sg.fmt = si == 0 ? q_fmt[0] : (si == 1 ? q_fmt[1] : q_fmt[2]);
    sg.dtype = si == 0 ? q_dt[0] : (si == 1 ? q_dt[1] : q_dt[2]);
    sg.L = si == 0 ? q_L[0] : (si == 1 ? q_L[1] : q_L[2]);
    sg.img_map = si == 0 ? q_map[0] : (si == 1 ? q_map[1] : q_map[2]);
    sg.pad_ = 0;
    return sg;
}

// ---------------- epilogue of an accumulated tile (m0, n0), shared by every main loop of this file ----------------
// acc[mi][ni]: the 16x16 fragments of the wave tile (WM x WN at wave position (wm, wn)); epi_smem: the workgroup's LDS, dead
// as a k-loop ring when this runs (the caller has NOT synchronised: the first thing the staged forms do is a barrier).
template <int BM, int BN, int WAVES_M, int WAVES_N, bool SWAP, int MINW>
__device__ __forceinline__ void igemm_epilogue(const IGemmArgs& e, f4 (&acc)[BM / WAVES_M / 16][BN / WAVES_N / 16], const int m0, const int n0,
                                               const int wm, const int wn, const int lane, const int wave, const int split, char* const epi_smem) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MI = WM / 16, NI = WN / 16;
    const half_t* Rptr = (const half_t*)e.res;
    if constexpr (SWAP) {
        // Operands were swapped (D = W.A^T): a lane owns ONE output row (lane&15) and FOUR consecutive columns
        // (lane>>4)*4+i of each 16x16 fragment.  Bias / time vector / GEGLU / SiLU are applied in registers; the
        // 16-row slab of the wave tile then goes through a wave-private LDS staging area (the k-loop ring is dead) and
        // leaves as whole 16-byte pieces of full output rows: residual reads and result writes are coalesced 128-byte+
        // row segments instead of 8-byte fragments scattered over 16 rows.
        __syncthreads();           // every wave is done with the ring
        // Mixed layouts (row-major segments + ONE transposed last segment: Q | K | V^T of a self-attention projection in one launch):
        // the dispatcher only picks tiles whose width divides the transposed segment's first column, so a tile lies in one segment
        // and the choice below is workgroup-uniform
        const bool t_last = e.nseg > 1 && n0 >= (e.nseg == 2 ? e.seg[1].col_begin : e.seg[2].col_begin);
        const int t_fmt_last = e.nseg == 1 ? e.seg[0].fmt : (e.nseg == 2 ? e.seg[1].fmt : e.seg[2].fmt);
        if ((e.nseg == 1 || t_last) && t_fmt_last == SEG_TRANSPOSED) {
            // One transposed segment (NCHW result / V^T operand): out[(img*ncols + c)*ld + tok].  Everything that is
            // indexed by (row, column) -- bias, time vector, SiLU, residual, scale -- is applied in the fragment layout;
            // a 16-channel slab of the wave tile (16 x WM tokens) is then transposed through the wave's LDS area and
            // leaves as 16-byte pieces of WM-token runs of one channel (256 B contiguous for WM = 128) instead of the
            // natural orientation's isolated 8-byte stores.
            constexpr int SLT = WM + 4;                    // 4*SLT = 16 (mod 64 banks): the 4 channel groups of a write hit distinct banks
            float* stg = (float*)epi_smem + wave * (16 * SLT);
            const IGemmSeg sg = seg_select(e, e.nseg - 1);
            const int erow = lane & 15, ecol = (lane >> 4) * 4;
            const int wrow0 = m0 + wm * WM;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int pcb = n0 + wn * WN + ni * 16;
                if (pcb < e.Nout) {
                    const int pcol = pcb + ecol;
                    f4 b4 = f4{0.f, 0.f, 0.f, 0.f};
                    if (e.bias) b4 = *(const f4*)(e.bias + pcol);
                    const float sc_t = col_scale(e, pcb);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const int row_w = wrow0 + mi * 16 + erow;
                        f4 x = acc[mi][ni] + b4;
                        if (row_w < e.M) {
                            if (e.rowvec) x += *(const f4*)(e.rowvec + (size_t)(row_w / e.rows_per_img) * e.rowvec_ld + pcol);
                            if (e.act == 1) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) x[i] = silu_f(x[i]);
                            }
                            if (Rptr) {
                                if (e.res_f32) {
                                    x += *(const f4*)((const float*)e.res + (size_t)row_w * e.ldres + pcol);
                                } else {
                                    const h4 rr = *(const h4*)(Rptr + (size_t)row_w * e.ldres + pcol);
#pragma unroll
                                    for (int i = 0; i < 4; ++i) x[i] += (float)rr[i];
                                }
                            }
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) stg[(ecol + i) * SLT + mi * 16 + erow] = x[i] * sc_t;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    constexpr int CH8 = WM / 8;
                    for (int idx = lane; idx < 16 * CH8; idx += 64) {
                        const int ch = idx / CH8, c8 = idx - ch * CH8;
                        const int row = wrow0 + c8 * 8;
                        if (row >= e.M) continue;
                        int img = row / sg.L;
                        const int tok = row - img * sg.L;
                        if (sg.img_map) img = sg.img_map[img];
                        const size_t o = ((size_t)img * sg.ncols + (pcb - sg.col_begin) + ch) * sg.ld + tok;
                        const f4 v0 = *(const f4*)(stg + ch * SLT + c8 * 8), v1 = *(const f4*)(stg + ch * SLT + c8 * 8 + 4);
                        if (sg.dtype == DT_F16) {
                            if (e.nonfinite && (out_of_half(v0[0]) || out_of_half(v0[1]) || out_of_half(v0[2]) || out_of_half(v0[3]) ||
                                                out_of_half(v1[0]) || out_of_half(v1[1]) || out_of_half(v1[2]) || out_of_half(v1[3])))
                                *(volatile int32_t*)e.nonfinite = 1;          // (divergent code: any lane may raise the flag)
                            h8 pk = {(half_t)v0[0], (half_t)v0[1], (half_t)v0[2], (half_t)v0[3],
                                     (half_t)v1[0], (half_t)v1[1], (half_t)v1[2], (half_t)v1[3]};
                            *(h8*)((half_t*)sg.out + o) = pk;
                        } else if (sg.dtype == DT_F32) {
                            *(f4*)((float*)sg.out + o) = v0;
                            *(f4*)((float*)sg.out + o + 4) = v1;
                        } else {
                            typedef u16 us8 __attribute__((ext_vector_type(8)));
                            us8 pk = {f32_to_bf16(v0[0]), f32_to_bf16(v0[1]), f32_to_bf16(v0[2]), f32_to_bf16(v0[3]),
                                      f32_to_bf16(v1[0]), f32_to_bf16(v1[1]), f32_to_bf16(v1[2]), f32_to_bf16(v1[3])};
                            *(us8*)((u16*)sg.out + o) = pk;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        } else {
            // Row-major outputs.  Per 16-row slab of the wave tile: bias / time vector / GEGLU / SiLU in the fragment layout, the slab
            // through a wave-private LDS area, then 8-column pieces of full rows: residual, scale, blend, mirror, store.  Everything a
            // piece needs that does not depend on the slab -- its row and column inside the slab (a division by a COMPILE-TIME piece
            // count: the run-time one cost ~40 VALU per piece), its segment, its scale, its byte offsets -- is computed once per tile;
            // with one workgroup per CU nothing overlaps this code, and its instruction count is the per-tile fixed cost of every
            // short-K GEMM of the path (round 4: 22 -> ~10 us per 256x256 tile).
            constexpr int OWMAX = WN;                          // output columns of the wave tile (half of it with GEGLU)
            constexpr int SLD = OWMAX + 4;                     // floats; +16 B keeps the b128 accesses conflict-light
            float* stg = (float*)epi_smem + wave * (16 * SLD);
            const int erow = lane & 15, ecol = (lane >> 4) * 4;
            const bool gg = e.geglu != 0;
            void* s1_out = e.seg[1].out; void* s2_out = e.seg[2].out;
            int64_t s1_ld = e.seg[1].ld, s2_ld = e.seg[2].ld;
            int s1_cb = e.seg[1].col_begin, s2_cb = e.seg[2].col_begin, s1_dt = e.seg[1].dtype, s2_dt = e.seg[2].dtype;
            asm volatile("" : "+s"(s1_out), "+s"(s2_out), "+s"(s1_ld), "+s"(s2_ld), "+s"(s1_cb), "+s"(s2_cb), "+s"(s1_dt), "+s"(s2_dt));
            const int wcol0 = gg ? ((n0 + wn * WN) >> 1) : (n0 + wn * WN);     // first output column of the wave tile
            const int nout_eff = gg ? (e.Nout >> 1) : e.Nout;
            const int wrow0 = m0 + wm * WM;
            // column-only terms, fetched once per tile (or per image): cv = bias + the per-image vector while the 16 rows of a slab
            // share an image (slabs start at multiples of 16); ONE register vector per fragment column -- the 80-wide wave tile
            // has no room for two
            const bool rv_uniform = MINW == 1 && e.rowvec && (e.rows_per_img % 16) == 0;      // (MINW > 1: the 128-register tiles spill with it)
            f4 cv[NI];
            auto load_cv = [&](int img) __attribute__((always_inline)) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int pcb = n0 + wn * WN + ni * 16;
                    cv[ni] = (e.bias && pcb < e.Nout) ? *(const f4*)(e.bias + pcb + ecol) : f4{0.f, 0.f, 0.f, 0.f};
                    if (rv_uniform && pcb < e.Nout && !(gg && (ni & 1))) cv[ni] += *(const f4*)(e.rowvec + (size_t)img * e.rowvec_ld + pcb + ecol);
                }
            };
            int img_cur = 0, img_rem = 0;
            if (rv_uniform) { img_cur = wrow0 / e.rows_per_img; img_rem = wrow0 - img_cur * e.rows_per_img; }
            int cv_img = img_cur;
            load_cv(img_cur);
            // the slab loop, for a compile-time number of 8-column pieces per staged row
            auto run = [&](auto owc_) __attribute__((always_inline)) {
                constexpr int OWC = decltype(owc_)::value;
                constexpr int RT = (16 * OWC + 63) / 64;               // pieces per lane per slab
                // residual pieces fetched AHEAD of their slab (depth RD), each refill issued before the stores of its own slab so that
                // waiting for it never means waiting for a store (one in-order counter): one slab ahead left every slab stalled
                // on a full memory latency with one workgroup per CU.  The 128-register two-workgroup tiles fetch at use.
                constexpr int RD = (MINW > 1 || WM * WN < 128 * 64 || NI > 4) ? 0 : 2;      // (the 80-wide wave tile has no registers to spare)
                // piece t of a lane: row pr(t) of the slab, output columns oc(t) .. + 8 (recomputed where used -- a shift or a
                // multiply-high by a constant -- rather than kept: the 128-register two-workgroup tiles have nothing to spare)
                auto pr = [&](int t) __attribute__((always_inline)) { return (lane + 64 * t) / OWC; };
                auto oc = [&](int t) __attribute__((always_inline)) { const int idx = lane + 64 * t; return wcol0 + (idx - (idx / OWC) * OWC) * 8; };
                auto pv = [&](int t) __attribute__((always_inline)) { return lane + 64 * t < 16 * OWC && oc(t) < nout_eff; };
                const bool has_res = Rptr != nullptr;
                f4 rf[RD ? RD : 1][RT][2];
                const ResSrc rs = {e.res, (unsigned)e.ldres, e.res_f32 != 0, e.res_up == 2, e.Hout * e.Wout, e.Wout, e.Hout >> 1, e.Wout >> 1};
                if (has_res) {
#pragma unroll
                    for (int d = 0; d < RD; ++d)
#pragma unroll
                        for (int t = 0; t < RT; ++t) {
                            rf[d][t][0] = rf[d][t][1] = f4{0.f, 0.f, 0.f, 0.f};
                            const int row = wrow0 + d * 16 + pr(t);
                            if (d < MI && pv(t) && row < e.M) res_fetch8(rs, row, oc(t), rf[d][t][0], rf[d][t][1]);
                        }
                }
                // fp16 rows with nothing to add after the transposition (projections, GEGLU, convolutions without a residual): the slab is
                // staged as PACKED fp16 -- scale and rounding happen in the fragment layout, the same fp32 operations in the same order --
                // so a piece is one 16-byte LDS read and one store: half the LDS traffic and a third of the per-piece instructions
                const bool one_row_seg = e.nseg == 1 || (e.nseg == 2 && e.seg[1].fmt == SEG_TRANSPOSED);      // (this tile then lies in segment 0)
                if (!has_res && !e.blend_mix && !e.out16 && one_row_seg && e.seg[0].dtype == DT_F16) {
                    constexpr int RS = OWMAX * 2 + 16;                 // bytes per staged row
                    char* const stg16 = (char*)stg;
                    half_t* const outp = (half_t*)e.seg[0].out;
                    const unsigned old_ = (unsigned)e.seg[0].ld;
                    const int ocb = e.seg[0].col_begin;
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const int row_w = wrow0 + mi * 16 + erow;
                        if (rv_uniform) {
                            if (img_cur != cv_img) { cv_img = img_cur; load_cv(img_cur); }
                            img_rem += 16;
                            if (img_rem >= e.rows_per_img) { img_rem -= e.rows_per_img; ++img_cur; }
                        }
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            if (gg && (ni & 1)) continue;
                            const int pcb = n0 + wn * WN + ni * 16;
                            if (pcb >= e.Nout) continue;
                            f4 x = acc[mi][ni] + cv[ni];
                            if (!rv_uniform && e.rowvec && row_w < e.M) x += *(const f4*)(e.rowvec + (size_t)(row_w / e.rows_per_img) * e.rowvec_ld + pcb + ecol);
                            if (gg) {
                                const f4 g = acc[mi][ni + (NI > 1 ? 1 : 0)] + cv[ni + 1 < NI ? ni + 1 : ni];
#pragma unroll
                                for (int i = 0; i < 4; ++i) x[i] *= gelu_erf_f(g[i]);
                            }
                            if (e.act == 1) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) x[i] = silu_f(x[i]);
                            }
                            const int lcol = (gg ? (ni >> 1) * 16 : ni * 16) + ecol;
                            const float sc = col_scale(e, wcol0 + lcol);
                            const h4 pk = {(half_t)(x[0] * sc), (half_t)(x[1] * sc), (half_t)(x[2] * sc), (half_t)(x[3] * sc)};
                            if (e.nonfinite) flag_nonfinite(e.nonfinite, row_w < e.M && (out_of_half(x[0] * sc) || out_of_half(x[1] * sc) || out_of_half(x[2] * sc) || out_of_half(x[3] * sc)));
                            *(h4*)(stg16 + erow * RS + lcol * 2) = pk;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                        for (int t = 0; t < RT; ++t) {
                            const int prt = pr(t);
                            const int row = wrow0 + mi * 16 + prt;
                            if (!pv(t) || row >= e.M) continue;
                            const int ocol = oc(t);
                            const h8 v = *(const h8*)(stg16 + prt * RS + (ocol - wcol0) * 2);
                            *(h8*)(outp + ((unsigned)row * old_ + (unsigned)(ocol - ocb))) = v;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    return;
                }
                const float al = e.blend_mix ? __builtin_amdgcn_rcpf(1.0f + __expf(-e.blend_mix[0])) : 0.f;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int row_w = wrow0 + mi * 16 + erow;
                    if (rv_uniform) {
                        if (img_cur != cv_img) { cv_img = img_cur; load_cv(img_cur); }      // (wave-uniform) the slab starts a new image
                        img_rem += 16;
                        if (img_rem >= e.rows_per_img) { img_rem -= e.rows_per_img; ++img_cur; }
                    }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        if (gg && (ni & 1)) continue;
                        const int pcb = n0 + wn * WN + ni * 16;
                        if (pcb >= e.Nout) continue;
                        f4 x = acc[mi][ni] + cv[ni];
                        if (!rv_uniform && e.rowvec && row_w < e.M) x += *(const f4*)(e.rowvec + (size_t)(row_w / e.rows_per_img) * e.rowvec_ld + pcb + ecol);
                        if (gg) {
                            const f4 g = acc[mi][ni + (NI > 1 ? 1 : 0)] + cv[ni + 1 < NI ? ni + 1 : ni];      // the gate's columns = the next fragment's
#pragma unroll
                            for (int i = 0; i < 4; ++i) x[i] *= gelu_erf_f(g[i]);
                        }
                        if (e.act == 1) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) x[i] = silu_f(x[i]);
                        }
                        const int lcol = (gg ? (ni >> 1) * 16 : ni * 16) + ecol;
                        *(f4*)(stg + erow * SLD + lcol) = x;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // same-wave LDS ops are in order; pin compiler order
#pragma unroll
                    for (int t = 0; t < RT; ++t) {
                        const int prt = pr(t);
                        const int row = wrow0 + mi * 16 + prt;
                        if (!pv(t) || row >= e.M) continue;
                        const int ocol = oc(t);
                        const int c8o = (ocol - wcol0);
                        const f4 v0 = *(const f4*)(stg + prt * SLD + c8o), v1 = *(const f4*)(stg + prt * SLD + c8o + 4);
                        float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                        if (has_res) {
                            f4 r0, r1;
                            if constexpr (RD > 0) {
                                r0 = rf[mi % RD][t][0];
                                r1 = rf[mi % RD][t][1];
                                if (mi + RD < MI && row + 16 * RD < e.M) res_fetch8(rs, row + 16 * RD, ocol, rf[mi % RD][t][0], rf[mi % RD][t][1]);     // refill: same piece, RD slabs on
                            } else {
                                res_fetch8(rs, row, ocol, r0, r1);
                            }
#pragma unroll
                            for (int i = 0; i < 4; ++i) { x[i] += r0[i]; x[4 + i] += r1[i]; }
                        }
                        {
                            // an 8-column chunk never straddles scale2_from (a multiple of 8, checked by op_igemm)
                            const float sc = col_scale(e, ocol);
#pragma unroll
                            for (int i = 0; i < 8; ++i) x[i] *= sc;
                        }
                        if (e.blend_mix) {   // AlphaBlender fold: (1-a) * this branch + a * the other branch
                            float bx[8];
                            if (e.blend_f32) {
                                const float* bp = (const float*)e.blend_x + (unsigned)((unsigned)row * (unsigned)e.ld_blend + (unsigned)ocol);
                                const f4 b0 = *(const f4*)bp, b1 = *(const f4*)(bp + 4);
#pragma unroll
                                for (int i = 0; i < 4; ++i) { bx[i] = b0[i]; bx[4 + i] = b1[i]; }
                            } else {
                                const h8 bb = *(const h8*)((const half_t*)e.blend_x + (unsigned)((unsigned)row * (unsigned)e.ld_blend + (unsigned)ocol));
#pragma unroll
                                for (int i = 0; i < 8; ++i) bx[i] = (float)bb[i];
                            }
#pragma unroll
                            for (int i = 0; i < 8; ++i) x[i] = al * bx[i] + (1.0f - al) * x[i];
                        }
                        if (e.nonfinite && (e.out16 != nullptr || e.seg[0].dtype == DT_F16)) {
                            bool bad = false;
#pragma unroll
                            for (int i = 0; i < 8; ++i) bad = bad || out_of_half(x[i]);
                            if (bad) *(volatile int32_t*)e.nonfinite = 1;     // (divergent code: any lane may raise the flag)
                        }
                        if (e.out16) {       // fp16 GEMM-operand mirror of an fp32 stream output
                            h8 pk;
#pragma unroll
                            for (int i = 0; i < 8; ++i) pk[i] = (half_t)x[i];
                            const unsigned o16 = (unsigned)row * (unsigned)e.ld16 + (unsigned)ocol;
                            *(h8*)((half_t*)e.out16 + o16) = pk;
                            if (e.out16_lo_off) {      // split operand: the rounding residual rides along (hi + lo == x to ~2^-22)
                                h8 lo;
#pragma unroll
                                for (int i = 0; i < 8; ++i) lo[i] = (half_t)(x[i] - (float)pk[i]);
                                *(h8*)((half_t*)e.out16 + e.out16_lo_off + o16) = lo;
                            }
                        }
                        // Segment of this 8-column piece (the last one whose first column is <= ocol): its fields selected VALUE by
                        // value from scalars (s1_* / s2_*, made opaque above: left to itself the compiler turns "select between two
                        // loaded kernel-argument fields" into "load from a selected address", which moves the whole descriptor to
                        // scratch for the entire kernel -- every argument read then waits on vmcnt)
                        const bool in1 = e.nseg > 1 && ocol >= s1_cb, in2 = e.nseg > 2 && ocol >= s2_cb;
                        void* const sg_out = in2 ? s2_out : (in1 ? s1_out : e.seg[0].out);
                        const int64_t sg_ld = in2 ? s2_ld : (in1 ? s1_ld : e.seg[0].ld);
                        const int sg_cb = in2 ? s2_cb : (in1 ? s1_cb : e.seg[0].col_begin);
                        const int sg_dt = in2 ? s2_dt : (in1 ? s1_dt : e.seg[0].dtype);
                        // 32-bit element offsets off the (scalar) bases: checked on the host (op_igemm: every row-major operand of the epilogue
                        // spans < 2^32 elements); 64-bit per-lane addresses cost the 80-wide wave tile its last registers
                        const unsigned o = (unsigned)row * (unsigned)sg_ld + (unsigned)(ocol - sg_cb);
                        if (sg_dt == DT_F16) {
                            h8 pk;
#pragma unroll
                            for (int i = 0; i < 8; ++i) pk[i] = (half_t)x[i];
                            *(h8*)((half_t*)sg_out + o) = pk;
                        } else if (sg_dt == DT_F32) {
                            float* const op = (float*)sg_out + (size_t)split * e.M * e.Nout;       // split > 0 only for fp32 slabs
                            *(f4*)(op + o) = f4{x[0], x[1], x[2], x[3]};
                            *(f4*)(op + o + 4) = f4{x[4], x[5], x[6], x[7]};
                        } else {
                            typedef u16 us8 __attribute__((ext_vector_type(8)));
                            us8 pk;
#pragma unroll
                            for (int i = 0; i < 8; ++i) pk[i] = f32_to_bf16(x[i]);
                            *(us8*)((u16*)sg_out + o) = pk;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // slab reads retired before the next slab is written
                }
            };
            if (gg) run(std::integral_constant<int, (OWMAX / 16 > 0 ? OWMAX / 16 : 1)>{});
            else run(std::integral_constant<int, OWMAX / 8>{});
        }
    } else {
        // C/D fragment map of v_mfma_f32_16x16x32: row = (lane>>4)*4 + i, col = lane&15
        const int erow = (lane >> 4) * 4, ecol = lane & 15;
#pragma clang loop unroll(full)      // (a hint alone left this loop rolled once the body grew: acc[][] indexed by a register = the accumulators in scratch)
        for (int ni = 0; ni < NI; ++ni) {
            if (e.geglu && (ni & 1)) continue;           // odd fragments are the gates of the even ones
            const int pcb = n0 + wn * WN + ni * 16;      // packed column block start
            if (pcb >= e.Nout) continue;
            const int pcol = pcb + ecol;                 // packed column (bias index)
            const int ocol = e.geglu ? ((pcb >> 5) << 4) + ecol : pcol;   // output column
            const int ocb = ocol - ecol;
            // segment lookup (segment boundaries are multiples of 16 -> uniform per fragment)
            int si = 0;
#pragma unroll
            for (int k = 1; k < 3; ++k)
                if (k < e.nseg && ocb >= e.seg[k].col_begin) si = k;
            const IGemmSeg sg = seg_select(e, si);
            const int scol = ocol - sg.col_begin;
            const float bh = e.bias ? e.bias[pcol] : 0.f;
            const float bg = (e.geglu && e.bias) ? e.bias[pcol + 16] : 0.f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int rbase = m0 + wm * WM + mi * 16 + erow;
                if (rbase >= e.M) continue;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = rbase + i;
                    float x = acc[mi][ni][i] + bh;
                    if (e.rowvec && row < e.M) x += e.rowvec[(size_t)(row / e.rows_per_img) * e.rowvec_ld + pcol];
                    if (e.geglu) {
                        const float g = acc[mi][ni + (NI > 1 ? 1 : 0)][i] + bg;
                        x = x * gelu_erf_f(g);
                    }
                    if (e.act == 1) x = silu_f(x);
                    if (Rptr && row < e.M)
                        x += e.res_f32 ? ((const float*)e.res)[res_row_of(e, row) * e.ldres + ocol] : (float)Rptr[res_row_of(e, row) * e.ldres + ocol];
                    v[i] = x * col_scale(e, ocol);
                }
                if (sg.fmt == SEG_ROW) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = rbase + i;
                        if (row < e.M) store_from_f32(sg.out, (size_t)row * sg.ld + scol, sg.dtype, v[i]);
                    }
                } else {
                    const int img = rbase / sg.L, tok = rbase - img * sg.L;
                    const int dimg = sg.img_map ? sg.img_map[img] : img;
                    const size_t base = ((size_t)dimg * sg.ncols + scol) * sg.ld + tok;
                    const bool vec = ((sg.L & 3) == 0) && ((sg.ld & 3) == 0) && (rbase + 3 < e.M);
                    if (vec) {
                        if (sg.dtype == DT_F16) {
                            h4 p = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                            *(h4*)((half_t*)sg.out + base) = p;
                        } else if (sg.dtype == DT_F32) {
                            *(f4*)((float*)sg.out + base) = f4{v[0], v[1], v[2], v[3]};
                        } else {
                            typedef u16 us4 __attribute__((ext_vector_type(4)));
                            us4 p = {f32_to_bf16(v[0]), f32_to_bf16(v[1]), f32_to_bf16(v[2]), f32_to_bf16(v[3])};
                            *(us4*)((u16*)sg.out + base) = p;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int row = rbase + i;
                            if (row >= e.M) break;
                            const int im = row / sg.L, tk = row - im * sg.L;
                            const int dm = sg.img_map ? sg.img_map[im] : im;
                            store_from_f32(sg.out, ((size_t)dm * sg.ncols + scol) * sg.ld + tk, sg.dtype, v[i]);
                        }
                    }
                }
            }
        }
    }

}

