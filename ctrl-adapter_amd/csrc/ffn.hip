// Fused GEGLU feed-forward of a transformer block with dim 512 / hidden 2048 on the CDNA4 matrix cores, gfx950 only (round 6):
//
//     out = stream + ( (X W1h^T + b1h) * gelu_erf(X W1g^T + b1g) ) W2^T + b2          X = LayerNorm(norm3)(stream), fp16 [M][512]
//
// -- diffusers FeedForward(GEGLU) of BasicTransformerBlock / TemporalBasicTransformerBlock (ff and ff_in), the way the adapter's
// spatial and temporal transformers use it (model/adapter_spatial_temporal.py:108-130, inner dim 512 for every pyramid level).  The
// two-launch form (igemm.hip: a GEGLU GEMM that writes the [M][2048] hidden activation, 537 MB at M = 131072, and a K = 2048 GEMM that
// reads it back) is 4.5 of 30 ms per SDXL step and 12.5 of 51 ms per SVD step; here the hidden activation never leaves the CU.
//
// One workgroup (8 waves as 2 (M) x 4 (N), two per SIMD) owns 128 token rows and walks the hidden units in 16 chunks of 128:
//
//   G1   S[128 x 256] = X[128 x 512] . W1c^T      (the chunk's 128 hidden + 128 gate rows of the interleaved GEGLU pack; 8 k-tiles of 64;
//                                                 wave tile 64 x 64 = two (hidden, gate) fragment pairs; accumulators start at b1)
//   GEGLU  P = S_h * gelu_erf(S_g) in registers, fp16, 4 bytes per lane and store into a 16 KB LDS area ([128 rows][2 k-steps x 32 hidden])
//   G2   O[128 x 512] += P[128 x 128] . W2c^T     (4 k-steps of 32 hidden; wave tile 64 x 128; O stays in 128 accumulator registers)
//
// and finishes with the implicit GEMM's own epilogue (igemm_epilogue.h: + b2, + fp32 / fp16 residual stream, fp32 master + fp16 mirror).
//
// Data movement.  X k-tiles, W1 and W2 tiles stream global -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane, masked rows = offsets
// beyond num_records: the hardware writes zeros) as 16 KB PIECES of [128 rows][64 halfs] (128-byte rows, 16-byte chunks XOR-swizzled on
// the source address and on the ds_read_b128 side: conflict-free, cdna_hip_programming.md rule 21): a G1 k-tile is three pieces (X, the
// W1 rows of wave columns 0-1, of 2-3), a G2 k-tile (64 hidden) four (128 output channels each, one per wave column); 32 pieces per
// chunk go round an 8-slot ring, so a piece's slot is a compile-time constant of its place in the chunk and every fragment read is
// `per-lane base + immediate`.  The X tile is re-streamed per chunk (128 KB from L2; the alternative -- X resident -- leaves no room for
// the weight ring in 160 KB), the weights are streamed once per workgroup: 98 FLOP per byte moved into LDS, against 128 for the 256 x 256
// GEMM tile.  b1 (16 KB fp32) is loaded once.
//
// Schedule (the 8-phase discipline of igemm8_kernel): the chunk body is 24 PHASES of 16 MFMAs per wave -- {fragment reads of the phase,
// LDS-DMA of later pieces, counted s_waitcnt vmcnt, lgkmcnt(0), barrier, MFMAs, barrier} -- and the two wave rows run one barrier apart,
// so one row's reads + DMA issue sit under the other row's MFMAs.  A piece is re-staged one phase after the phase that read it last
// (every wave retires its reads BEFORE the phase's first barrier), waited for one phase before its first read (the other wave row's
// barrier lies in between), and the DMA queue is never drained except once per chunk.  Half of a chunk's GEGLU math (k-steps 2, 3) is
// interleaved with the MFMAs of k-steps 0, 1; the first half is exposed (it is what frees the registers the interleaved half needs).
#include "ops.h"
#include <cstdio>
#include <cstring>
#include <type_traits>

namespace {

#include "igemm_epilogue.h"

struct FfnGroup { FfnArgs a[kMaxGroup]; };
#define FFN_GROUP_ARGS(gp) (((const FfnArgs*)__builtin_amdgcn_kernarg_segment_ptr())[gp])

namespace ff {
constexpr int D = 512, H = 2048, HC = 128, NCH = H / HC;       // dim, hidden, hidden units per chunk, chunks
constexpr int PIECE = 16384;                                   // [128 rows][64 halfs]
constexpr int P_BASE = 8 * PIECE;                              // P: [128 rows][2 slots x 32 hidden] fp16
constexpr int B1_BASE = P_BASE + PIECE;                        // b1: 4096 fp32
constexpr int LDS_TOTAL = B1_BASE + 2 * H * 4;                 // 163840
constexpr unsigned OOB = 0x80000000u;
__host__ __device__ constexpr int slot(int piece) { return (piece & 7) * PIECE; }
}  // namespace ff

#define FF_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// end of a phase's load segment: this wave's fragment reads are retired BEFORE the barrier (so a piece may be re-staged one phase after
// its last read), then the barrier that opens the MFMA segment
#define FF_PRE()                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_setprio(1);
#define FF_POST()                                          \
    __builtin_amdgcn_s_setprio(0);                         \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    __builtin_amdgcn_sched_barrier(0);
// the same with the P stores of an interleaved GEGLU unit retired before the barrier that lets the next k-step read them
#define FF_POST_P()                                        \
    __builtin_amdgcn_s_setprio(0);                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    __builtin_amdgcn_sched_barrier(0);

// JIT (tests only, CTRL_FF_FUSED=jitter): every wave sleeps a pseudo-random 0 .. 31 x 64 cycles at the head of both segments of every phase, a
// different amount per wave, phase and chunk -- the waves of a workgroup then arrive at their barriers in every order, which is what turns a
// missing wait or a too-early re-staging into wrong numbers instead of a coincidence (a co-resident kernel of another stream lane does the
// same to the product kernel, rarely).  The result must be bit-identical to the plain kernel's.
// Ablation builds (tools/bin/ffn_bench only; WRONG results by construction): ABL_NODMA stages nothing inside the chunk loop, ABL_NOGEGLU replaces
// the GEGLU math by a plain product, ABL_NOREAD skips the fragment reads -- what each costs is the time that build does NOT take.
enum { FF_PLAIN = 0, FF_JITTER = 1, ABL_NODMA = 2, ABL_NOGEGLU = 3, ABL_NOREAD = 4, FF_BULK = 5, ABL_XHOT = 6 };
template <int MODE>
__global__ __launch_bounds__(512, 2) void ffn512_kernel(FfnGroup kargs, int per, int ntm) {
    constexpr bool JIT = MODE == FF_JITTER;
    // (measured and dropped, profiles/r06_ffn_variants.txt: the LDS-DMA issued between the MFMAs instead of in the load segment, 0.93-0.95 vs
    // 0.90-0.92 ms at M = 131072; only a quarter of the GEGLU exposed with the rest one k-step ahead of its MFMAs -- all of S live through
    // k-step 0 --, 0.97-0.98 ms.)  FF_BULK: the first schedule (a k-tile's three pieces issued together), kept for A/B runs.
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int gp = __builtin_amdgcn_readfirstlane(blockIdx.x / per);      // grouped launch: see igemm_kernel
    const int gbid = blockIdx.x - gp * per;
    if (gbid >= ntm) return;
    const FfnArgs& a = FFN_GROUP_ARGS(gp);
    const IGemmArgs& e = a.out;
    typedef void __attribute__((address_space(3)))* lptr_t;
    typedef const h8 __attribute__((address_space(3)))* lds_h8_t;
    typedef const f4 __attribute__((address_space(3)))* lds_f4_t;
    typedef h2 __attribute__((address_space(3)))* lds_h2_t;
#define FF_LDS_DST(off) ((lptr_t)(size_t)(unsigned)(off))

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int m0 = gbid * 128;
    unsigned jit_state = 0x9E3779B9u * (unsigned)(wave + 1) + (unsigned)gbid * 0x85EBCA6Bu;
    auto jitter = [&]() __attribute__((always_inline)) {
        if constexpr (JIT) {
            jit_state = jit_state * 1664525u + 1013904223u;
            const int n = (int)(__builtin_amdgcn_readfirstlane(jit_state) >> 27);
            for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
        }
    };

    // ---- staging coordinates.  Pass i of a piece: this wave writes LDS rows i*64 + wave*8 + lane/8, 16-byte chunk lane%8; the chunk
    //      swizzle of LDS row r is (r >> 1) & 7 = ((wave & 1) * 4 + (lane / 16)) & 7 for both passes ----
    const int lrow = lane >> 3, lpos = lane & 7;
    const int cgb = (lpos ^ ((((wave & 1) << 2) + (lrow >> 1)) & 7)) * 16;        // swizzled source chunk, bytes
    // ONE per-lane byte offset serves all three operands (registers are what this kernel is short of): every operand has 1024-byte
    // rows -- X [M][512] dense (checked on the host), W1 [4096][512], W2 re-packed as four column blocks [4][512][512] -- so row
    // (wave * 8 + lane / 8) of a pass is the same offset everywhere; pass 1 = + 64 rows, added at the use.  The X descriptor starts at
    // this workgroup's first row and ends at row M: rows beyond M are out of range, the hardware writes zeros.
    const unsigned vrow = (unsigned)((wave * 8 + lrow) * 1024 + cgb);
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((char*)const_cast<void*>(a.X) + (size_t)m0 * 1024, 0, (e.M - m0) * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W1), 0, (int)0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W2p), 0, (int)0x80000000u, 0x00020000);

    // piece I of chunk cc -> its ring slot.  Pieces of a chunk, in stream order: G1 k-tile t (t = 0..7): 3t = X[:, 64t..], 3t+1 = W1 rows
    // 0..127 of the chunk, 3t+2 = rows 128..255;  G2 k-tile f (f = 0, 1): 24 + 4f + q = W2 output channels 128q.., hidden 64f.. of the chunk
    auto issue = [&](auto I_, const int cc) __attribute__((always_inline)) {
        constexpr int I = decltype(I_)::value;
        const int dst = ff::slot(I) + wave * 1024;
        unsigned v0 = vrow;
        asm volatile("" : "+v"(v0));                      // (kept opaque: the + 64 rows below are one add at the use, not a second live register)
        const unsigned v1 = v0 + 64 * 1024;
        if constexpr (I < 24) {
            constexpr int t = I / 3, r = I % 3;
            if constexpr (r == 0) {
                // (ABL_XHOT: every X piece re-reads k-tile 0 -- 16 KB per workgroup, always in the L2: what the re-streamed X tile costs in the memory system)
                const int xo = (MODE == ABL_XHOT) ? 0 : t * 128;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, FF_LDS_DST(dst), 16, v0, xo, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, FF_LDS_DST(dst + 8192), 16, v1, xo, 0, 0);
            } else {
                const int so = (cc * 256 + (r - 1) * 128) * 1024 + t * 128;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w1_rs, FF_LDS_DST(dst), 16, v0, so, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w1_rs, FF_LDS_DST(dst + 8192), 16, v1, so, 0, 0);
            }
        } else {
            // W2 pack: [column block cc / 4][512 output channels][512]: hidden 128 cc + 64 f of the chunk = column (cc % 4) * 128 + 64 f of the block
            constexpr int f = (I - 24) / 4, q = (I - 24) % 4;
            const int so = ((cc >> 2) * 512 + q * 128) * 1024 + ((cc & 3) * 128 + f * 64) * 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w2_rs, FF_LDS_DST(dst), 16, v0, so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w2_rs, FF_LDS_DST(dst + 8192), 16, v1, so, 0, 0);
        }
    };
#define FF_ISSUE(I, cc) issue(std::integral_constant<int, (I)>{}, (cc))
#define FF_ISSUE_L(I, cc) do { if constexpr (MODE != ABL_NODMA) issue(std::integral_constant<int, (I)>{}, (cc)); } while (0)

    // ---- fragment read coordinates: row (lane & 15) of a 16-row fragment, logical 16-byte chunk (lane >> 4) [+ 4 for the second k-step
    //      = address ^ 64]; everything else is an immediate ----
    const int frow = lane & 15, fq = lane >> 4;
    const int sw = (frow >> 1) & 7;
    // ONE per-lane read base (row frow of a fragment, swizzled chunk fq); what depends on the wave only is added as a scalar at the use:
    //   X   + wm * 8192 + slot of the X piece + mi * 2048                 W1  + (wn & 1) * 8192 + slot of the wave column's W1 piece + ni * 2048
    //   W2  + wn * PIECE + (k-tile & 1) * 4 slots + ni * 2048             P   + P_BASE + wm * 8192 + mi * 2048; second k-step / P slot = ^ 64
    const int rbase = frow * 128 + ((fq ^ sw) << 4);
    const int pw0 = ff::P_BASE + (wm * 64 + frow) * 128 + ((wn ^ sw) << 4) + fq * 4;      // P writes (4 bytes): chunk = writer's wave column
    const int sx = wm * 8192, sw1 = (wn & 1) * 8192, sw2 = wn * ff::PIECE, sp = ff::P_BASE + wm * 8192;

    f4 sacc[4][4];      // S^T fragments: [token fragment mi][ni: 0 = hidden 0-15, 1 = their gates, 2 = hidden 16-31, 3 = their gates] of the wave column
    f4 oacc[4][8];      // O^T fragments: [mi][ni: output channels wn * 128 + 16 ni ..]
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) oacc[mi][ni] = f4{0.f, 0.f, 0.f, 0.f};
    h8 xf[4], wf[4], pf[4];
    if constexpr (MODE == ABL_NOREAD) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { xf[i] = h8{1, 1, 1, 1, 1, 1, 1, 1}; wf[i] = xf[i]; pf[i] = xf[i]; }
    }

    auto s_init = [&](const int cc) __attribute__((always_inline)) {      // S = b1 (the bias rides in the accumulator)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const f4 b = *(lds_f4_t)(size_t)(unsigned)((lane & 0x30) + ff::B1_BASE + wn * 256 + cc * 1024 + ni * 64);      // this lane's 4 packed columns of fragment ni
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) sacc[mi][ni] = b;
        }
    };
    // GEGLU of k-step KAP (hidden (KAP >> 1) * 16 + 4 fq + 2 (KAP & 1) + {0, 1} of the wave column's 32), token fragment MI: two products,
    // packed fp16, one 4-byte store.  k-step KAP of G2 then reads, per row, chunk (KAP & 1) * 4 + wave column = 8 hidden of that wave column
    float pmax = 0.f;                 // range check of the fp16 hidden activation: branch-free in the loop, one test at the end
#define FF_GEGLU(KAP, MI)                                                                                              \
    do {                                                                                                               \
        float p0_ = sacc[MI][2 * ((KAP) >> 1)][2 * ((KAP) & 1)], p1_ = sacc[MI][2 * ((KAP) >> 1)][2 * ((KAP) & 1) + 1];                      \
        if constexpr (MODE == ABL_NOGEGLU) {                                                                          \
            p0_ *= sacc[MI][2 * ((KAP) >> 1) + 1][2 * ((KAP) & 1)]; p1_ *= sacc[MI][2 * ((KAP) >> 1) + 1][2 * ((KAP) & 1) + 1];                 \
        } else {                                                                                                       \
            p0_ *= gelu_erf_f(sacc[MI][2 * ((KAP) >> 1) + 1][2 * ((KAP) & 1)]);                                         \
            p1_ *= gelu_erf_f(sacc[MI][2 * ((KAP) >> 1) + 1][2 * ((KAP) & 1) + 1]);                                     \
        }                                                                                                              \
        /* (in place, one instruction: left to the compiler the maximum chain sinks to the end of the chunk body and keeps every    \
           product alive -- in scratch -- until there) */                                                                  \
        asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(pmax) : "v"(p0_), "v"(p1_));                                \
        const h2 pk_ = {(half_t)p0_, (half_t)p1_};                                                                      \
        int pwb_ = pw0;                                                                                                \
        asm volatile("" : "+v"(pwb_));                                                                                 \
        *(lds_h2_t)(size_t)(unsigned)((((KAP) & 1) ? (pwb_ ^ 64) : pwb_) + (MI) * 2048) = pk_;                          \
    } while (0)

    // ---- prologue: b1, then the first 8 pieces; wait for b1 + k-tile 0 ----
    {
        const __amdgpu_buffer_rsrc_t b1_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.b1), 0, 2 * ff::H * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b1_rs, FF_LDS_DST(ff::B1_BASE + (i * 8 + wave) * 1024), 16, (unsigned)(lane * 16), (i * 8 + wave) * 1024, 0, 0);
    }
    FF_ISSUE(0, 0); FF_ISSUE(1, 0); FF_ISSUE(2, 0); FF_ISSUE(3, 0); FF_ISSUE(4, 0); FF_ISSUE(5, 0); FF_ISSUE(6, 0); FF_ISSUE(7, 0);
    FF_VMCNT(10);
    __builtin_amdgcn_s_barrier();
    s_init(0);
    if (wm == 1) __builtin_amdgcn_s_barrier();          // the second wave row runs one barrier behind

    // one G1 phase: k-step KK of k-tile T
#define FF_G1_READ(T, KK)                                                                                                       \
    {                                                                                                                           \
        jitter();                                                                                                               \
        /* (the per-lane bases are made opaque at every use: left alone, hipcc hoists all 40 `base + slot` sums of the chunk body   \
           out of the chunk loop and spills them -- scratch reloads inside the loop, which the in-order vmcnt turns into drains     \
           of the DMA queue) */                                                                                                 \
        int rb_ = rbase;                                                                                                        \
        asm volatile("" : "+v"(rb_));                                                                                           \
        if (KK) rb_ ^= 64;                                                                                                      \
        const int xb_ = rb_ + (sx + ff::slot(3 * (T)));                                                                         \
        const int wb_ = rb_ + (sw1 + ((wn >> 1) ? ff::slot(3 * (T) + 2) : ff::slot(3 * (T) + 1)));                              \
        if constexpr (MODE != ABL_NOREAD) {                                                                                     \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) xf[mi] = *(lds_h8_t)(size_t)(unsigned)(xb_ + mi * 2048);              \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) wf[ni] = *(lds_h8_t)(size_t)(unsigned)(wb_ + ni * 2048);              \
        } else { asm volatile("" : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]), "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]) : "v"(xb_), "v"(wb_)); }   \
    }
#define FF_G1_MMA()                                                                                                             \
    jitter();                                                                                                                   \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                                            \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                                        \
            sacc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], sacc[mi][ni], 0, 0, 0);
    // one G2 phase: k-step KAP (k-tile KAP >> 1, half KAP & 1), output fragments 4 NH .. 4 NH + 3
#define FF_G2_READ(KAP, NH)                                                                                                     \
    {                                                                                                                           \
        jitter();                                                                                                               \
        int rb_ = rbase;                                                                                                        \
        asm volatile("" : "+v"(rb_));                                                                                           \
        if ((KAP) & 1) rb_ ^= 64;                                                                                               \
        const int pb_ = rb_ + sp;                                                                                               \
        const int wb_ = rb_ + (sw2 + ((KAP) >> 1) * 4 * ff::PIECE + (NH) * 4 * 2048);                                           \
        if constexpr (MODE != ABL_NOREAD) {                                                                                     \
        if ((NH) == 0) { _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) pf[mi] = *(lds_h8_t)(size_t)(unsigned)(pb_ + mi * 2048); }   \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) wf[ni] = *(lds_h8_t)(size_t)(unsigned)(wb_ + ni * 2048);              \
        } else { asm volatile("" : "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]), "+v"(pf[3]), "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]) : "v"(pb_), "v"(wb_)); }   \
    }
#define FF_G2_MMA(NH, MI)                                                                                                       \
    if ((MI) == 0) jitter();                                                                                                    \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                                            \
        oacc[MI][(NH) * 4 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], pf[MI], oacc[MI][(NH) * 4 + ni], 0, 0, 0);

#pragma clang loop unroll(disable)
    for (int c = 0; c < ff::NCH; ++c) {
        const bool more = c + 1 < ff::NCH;
        // the 24 phases of a chunk: generated (and hazard-checked) by tools/gen_ffn_schedule.py from the schedule named in the file name
        if constexpr (MODE == FF_BULK) {
#include "ffn_body_bulk.inc"
        } else {
#include "ffn_body_spread.inc"
        }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    if (e.nonfinite) flag_nonfinite(e.nonfinite, out_of_half(pmax));

    // the lane index is re-derived HERE, by an instruction the compiler cannot move: everything the epilogue computes from it (rows, columns,
    // byte offsets -- all loop-invariant) was otherwise hoisted above the chunk loop and held 9 registers across it, which the loop does not have
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    igemm_epilogue<128, 512, 2, 4, true, 1>(e, oacc, m0, 0, wm, wn, lane_e, wave, 0, smem_raw);
}

// W2 [512][2048] (the linear pack: K contiguous) -> the layout and k order of the fused kernel ([4][512][512], see the kernel): inside every chunk of 128 hidden units, k-step kap
// (32 units), 16-byte chunk q' (8 units, written by wave column q'), element 2 q + i holds hidden unit 32 q' + 16 (kap >> 1) + 4 q +
// 2 (kap & 1) + i of the chunk -- the unit whose GEGLU product lane group q of wave column q' stores there (FF_GEGLU)
__global__ __launch_bounds__(256) void ffn_pack_w2_kernel(const half_t* __restrict__ w, half_t* __restrict__ out, int N, int K) {
    const size_t total = (size_t)N * K;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t n = idx / K;
        const int k = (int)(idx - n * K);
        const int c = k >> 7, kk = k & 127;
        const int kap = kk >> 5, qp = (kk >> 3) & 3, el = kk & 7, q = el >> 1, i = el & 1;
        const int u = 32 * qp + 16 * (kap >> 1) + 4 * q + 2 * (kap & 1) + i;
        // four column blocks of 512 (1024-byte rows, like X and W1: one staging offset serves all operands): out[c / 4][n][(c % 4) * 128 + kk]
        out[((size_t)(c >> 2) * N + n) * 512 + (c & 3) * 128 + kk] = w[n * K + c * 128 + u];
    }
}

}  // namespace

bool op_ffn_fused_shape_ok(int dim, int inner) { return dim == ff::D && inner == ff::H; }

int op_ffn_pack_w2(const half_t* w2, half_t* out, int N, int K, hipStream_t s) {
    CTRL_CHECK(N == ff::D && K == ff::H, "ffn_pack_w2: the fused feed-forward is built for 2048 -> 512");
    LAUNCH("pack", ffn_pack_w2_kernel, dim3(1024), dim3(256), 0, s, w2, out, N, K);
    return 0;
}

static int ffn_check(const FfnArgs& a) {
    const IGemmArgs& e = a.out;
    CTRL_CHECK(a.X && a.W1 && a.b1 && a.W2p, "ffn: null operand");
    CTRL_CHECK(e.Nout == ff::D && e.M > 0, "ffn: the output GEMM must be [M][512]");
    CTRL_CHECK(a.ldx == ff::D && (((uintptr_t)a.X | (uintptr_t)a.W1 | (uintptr_t)a.W2p | (uintptr_t)a.b1) & 15) == 0, "ffn: operands must be 16-byte aligned");
    CTRL_CHECK((double)e.M * (double)a.ldx * 2.0 < 2147483648.0, "ffn: X beyond 2 GiB");
    CTRL_CHECK(e.nseg == 1 && e.seg[0].fmt == SEG_ROW && e.seg[0].col_begin == 0 && e.seg[0].ncols == ff::D && !e.geglu && !e.rowvec && !e.act &&
               e.res_up != 2 && e.scale2_from == 0, "ffn: one row-major output segment, no GEGLU / per-image vector / activation on the output GEMM");
    // the epilogue addresses rows with 32-bit element offsets
    CTRL_CHECK((double)e.M * (double)e.seg[0].ld < 4294967296.0 && (!e.res || (double)e.M * (double)e.ldres < 4294967296.0) &&
               (!e.out16 || (double)e.M * (double)e.ld16 < 4294967296.0), "ffn: output beyond 2^32 elements");
    return 0;
}

int op_ffn_fused_group(const FfnArgs* as, int n, hipStream_t s) {
    CTRL_CHECK(n >= 1 && n <= kMaxGroup, "ffn: group size");
    FfnGroup g;
    for (int i = 0; i < n; ++i) {
        TRY(ffn_check(as[i]));
        CTRL_CHECK(as[i].out.M == as[0].out.M, "ffn: the problems of a group must have the same M");
        g.a[i] = as[i];
        if (range_check_on() && !g.a[i].out.nonfinite) g.a[i].out.nonfinite = range_flag();
    }
    for (int i = n; i < kMaxGroup; ++i) g.a[i] = as[0];
    const int M = as[0].out.M;
    const int ntm = (M + 127) / 128;
    const int per = n > 1 ? ((ntm + 7) & ~7) : ntm;
    static bool attr_done[kMaxDevices] = {};
    const int dev = cur_device();
    if (!attr_done[dev]) {
        HIP_TRY(hipFuncSetAttribute((const void*)ffn512_kernel<FF_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, ff::LDS_TOTAL));
        HIP_TRY(hipFuncSetAttribute((const void*)ffn512_kernel<FF_JITTER>, hipFuncAttributeMaxDynamicSharedMemorySize, ff::LDS_TOTAL));
        HIP_TRY(hipFuncSetAttribute((const void*)ffn512_kernel<ABL_NODMA>, hipFuncAttributeMaxDynamicSharedMemorySize, ff::LDS_TOTAL));
        HIP_TRY(hipFuncSetAttribute((const void*)ffn512_kernel<ABL_NOGEGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, ff::LDS_TOTAL));
        HIP_TRY(hipFuncSetAttribute((const void*)ffn512_kernel<ABL_NOREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, ff::LDS_TOTAL));
        HIP_TRY(hipFuncSetAttribute((const void*)ffn512_kernel<FF_BULK>, hipFuncAttributeMaxDynamicSharedMemorySize, ff::LDS_TOTAL));
        HIP_TRY(hipFuncSetAttribute((const void*)ffn512_kernel<ABL_XHOT>, hipFuncAttributeMaxDynamicSharedMemorySize, ff::LDS_TOTAL));
        attr_done[dev] = true;
    }
    PROF_WORK(n * 2.0 * M * (double)(2 * ff::H * ff::D + ff::H * ff::D),
              n * ((double)M * ff::D * 2.0 + 3.0 * ff::H * ff::D * 2.0 + (double)M * ff::D * (as[0].out.seg[0].dtype == DT_F32 ? 4.0 : 2.0) +
                   (as[0].out.res ? (double)M * ff::D * (as[0].out.res_f32 ? 4.0 : 2.0) : 0.0) + (as[0].out.out16 ? (double)M * ff::D * 2.0 : 0.0)));
    char grp[16] = "";
    if (n > 1) snprintf(grp, sizeof(grp), " x%d", n);
    prof_detail("M%d dim512 hidden2048 geglu fused%s", M, grp);
    const char* pol = policy_raw(P_FF_FUSED);
    const dim3 grid((unsigned)(per * n)), blk(512);
    if (pol && pol[0] == 'j') LAUNCH("ffn_fused", (ffn512_kernel<FF_JITTER>), grid, blk, ff::LDS_TOTAL, s, g, per, ntm);
    else if (pol && !strcmp(pol, "abl_nodma")) LAUNCH("ffn_fused", (ffn512_kernel<ABL_NODMA>), grid, blk, ff::LDS_TOTAL, s, g, per, ntm);
    else if (pol && !strcmp(pol, "abl_nogeglu")) LAUNCH("ffn_fused", (ffn512_kernel<ABL_NOGEGLU>), grid, blk, ff::LDS_TOTAL, s, g, per, ntm);
    else if (pol && !strcmp(pol, "abl_xhot")) LAUNCH("ffn_fused", (ffn512_kernel<ABL_XHOT>), grid, blk, ff::LDS_TOTAL, s, g, per, ntm);
    else if (pol && !strcmp(pol, "bulk")) LAUNCH("ffn_fused", (ffn512_kernel<FF_BULK>), grid, blk, ff::LDS_TOTAL, s, g, per, ntm);
    else if (pol && !strcmp(pol, "abl_noread")) LAUNCH("ffn_fused", (ffn512_kernel<ABL_NOREAD>), grid, blk, ff::LDS_TOTAL, s, g, per, ntm);
    else LAUNCH("ffn_fused", (ffn512_kernel<FF_PLAIN>), grid, blk, ff::LDS_TOTAL, s, g, per, ntm);
    return 0;
}

int op_ffn_fused(const FfnArgs& a, hipStream_t s) {
    if (t_collect) {                      // grouped launches: deposit, the collector's flush launches the siblings together
        int rc = 0;
        const int i = t_collect->slot(OpCollector::FFN, s, &rc);
        if (rc) return rc;
        t_collect->ff[i] = a;
        return 0;
    }
    return op_ffn_fused_group(&a, 1, s);
}
