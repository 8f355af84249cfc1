// Implicit-GEMM on the CDNA4 matrix cores (v_mfma_f32_16x16x32_f16), gfx950 only.
//
// One kernel family covers every dense contraction of the hot path:
//   * Linear / 1x1 conv            (IG_ROWS)      -- reference: every nn.Linear / Conv2d(k=1) under
//                                                     controlnet/controlnet.py:366-424 and
//                                                     model/adapter_spatial_temporal.py:56-69
//   * 3x3 conv, stride 1|2, pad 1, optional nearest x2 up-sampling folded into the gather
//                                   (IG_CONV2D)    -- model/resnet_block_2d.py:164-221 (conv1/conv2,
//                                                     Upsample2D), diffusers Downsample2D
//   * Conv3d (3,1,1) over frames    (IG_TEMPORAL)  -- TemporalResnetBlock, adapter_spatial_temporal.py:96-104
//
// Layout: activations are channels-last fp16 ([pixels][C]); the K axis of the GEMM is (tap, cin) with
// cin contiguous, so every A-tile row is one 16-byte-vectorisable run of channels of one (shifted)
// input pixel; zero padding is a predicated load.  Weights are pre-packed [Cout][tap][Cin].
//
// Tiling: BM x BN x BK per workgroup (4 or 8 wavefronts), A/B tiles staged global -> LDS with the asynchronous
// LDS-DMA (global_load_lds_dwordx4: no VGPR round trip) into an NSTAGE-deep ring: NSTAGE-1 tiles in flight,
// a counted s_waitcnt vmcnt(N) + raw s_barrier per k-tile so later tiles stay in flight across the barrier; LDS tiles are linear and bank conflicts of the ds_read_b128 fragment reads are removed
// by an XOR chunk swizzle applied on the SOURCE address and on the read (MI355X_MICROARCH.md, LDS section;
// cdna_hip_programming.md rule 21).  Zero padding (conv halo, ragged M/N) = loading from a zero page.  fp32 accumulation; the epilogue
// fuses bias, per-image channel vector (time embedding), GEGLU, residual add, scale and the output
// layout (row-major slice, or transposed [img][C][tokens] for NCHW results / the V^T attention operand).
#include "ops.h"
#include "tile_order.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace {

// Grouped launch (round 5): up to kMaxGroup problems of the SAME shape and epilogue form -- the sibling adapter blocks of one
// pyramid level (model/ctrl_adapter.py:181-191 runs them one after the other; they are independent) -- share one launch: the
// descriptors travel as an array in the kernel-argument segment, workgroup b works on problem b / per.  A 32^2 / 16^2 GEMM then
// has 2-4 x the tiles (the wide tile instead of the 128 x 128 / 64 x 64 ring tiles) and the step has fewer launches.  A plain launch
// is a group of one.  Given the tile, every problem is computed exactly as it would be alone (same k order); the DISPATCHER, though, sizes the
// tile for the whole group by default (ctrl_group_launches(1): another tile family for some problems, last-bit differences) -- with
// ctrl_group_launches(2) it picks the tile a lone problem would get and the results are bit-identical to one-by-one launches.
struct IGemmGroup { IGemmArgs a[kMaxIGemmGroup]; };
// descriptor of this workgroup's problem, read from the kernel-argument segment (constant memory: scalar loads; the group is the first
// kernel parameter, i.e. offset 0 of the segment).  Not `kargs.a[gp]`: a dynamic index into the by-value parameter would pin its
// private copy -- 1.5 KB per lane -- in scratch.
#define IGEMM_GROUP_ARGS(gp) (((const IGemmArgs*)__builtin_amdgcn_kernarg_segment_ptr())[gp])

#include "igemm_epilogue.h"

// LDS tiles are linear [rows][BK] (global_load_lds writes wave-uniform base + lane*16), so bank conflicts of the
// ds_read_b128 fragment reads are removed by an XOR swizzle of the 16-byte chunk index that is applied to the
// per-lane SOURCE address when staging and to the read address when fetching fragments (same involution).
template <int BK>
__device__ __forceinline__ int chunk_swz(int row) {
    if (BK == 64) return (row >> 1) & 7;          // 128-B rows: 2 rows per 256-B bank line
    return (0 - (row >> 2)) & 3;                   // 64-B rows: 4 rows per bank line; {0,3,2,1} keeps mixed-chunk groups apart
}

// 128x256 / 256x128 tiles of 8 waves keep two workgroups resident per CU (72 KiB of LDS ring, <= 128 VGPRs per wave): one
// workgroup's epilogue (HBM-bound fp32 stream traffic, GEGLU math) overlaps the other one's k-loop
template <int BM, int BN, int NW> struct MinWaves { static constexpr int v = (BM * BN == 128 * 256) ? (NW == 8 ? 4 : (NW == 4 ? 2 : 1)) : 1; };

// (A persistent-workgroup form -- one workgroup per CU walking its tiles with one LDS ring running across them -- was built
// in round 2 and measured again at the start of round 3 after the epilogue fetch hoisting: 0.72-1.03x the one-tile form on
// every shape of the path, profiles/r03_gemm_persistent_form.txt.  It is gone from the product.)
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int NSTAGE, int MODE, bool SWAP>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, (MinWaves<BM, BN, WAVES_M * WAVES_N>::v)) void igemm_kernel(IGemmGroup kargs, int per, int ntm, int ntn, const half_t* zeros, int splitk, int order) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // grouped launch: workgroups [gp * per, (gp + 1) * per) belong to problem gp (per = the problem's workgroup count, rounded up to
    // a multiple of 8 in a group so that the XCD of a workgroup is still its local index % 8; the padding workgroups leave here)
    const int gp = __builtin_amdgcn_readfirstlane(blockIdx.x / per);      // (the division runs on the vector ALU: back to a scalar, so that
    const int gbid = blockIdx.x - gp * per;                                //  the descriptor below is read with scalar loads)
    if (gbid >= ntm * ntn * splitk) return;
    const IGemmArgs& a = IGEMM_GROUP_ARGS(gp);
    constexpr int NT = WAVES_M * WAVES_N * 64, NW = NT / 64;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;   // per-wave tile
    constexpr int MI = WM / 16, NI = WN / 16;
    constexpr int CPR = BK / 8;              // 16-B chunks per tile row
    constexpr int RPW = 64 / CPR;            // rows written by one wave-wide global_load_lds (1 KiB)
    constexpr int APASS = BM / (NW * RPW);
    constexpr int BPASS = (BN + NW * RPW - 1) / (NW * RPW);       // BN = 320: the last pass is half empty ...
    constexpr int BNP = BPASS * NW * RPW;                          // ... so the LDS B tile is padded to whole passes
    static_assert(APASS >= 1 && BM % (NW * RPW) == 0, "tile too small for the workgroup");
    static_assert(WN % 16 == 0 && WM % 16 == 0, "wave tile must be a multiple of the 16x16 fragment");

    half_t* As = (half_t*)smem_raw;                  // [NSTAGE][BM][BK]
    half_t* Bs = As + NSTAGE * BM * BK;              // [NSTAGE][BNP][BK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // XCD-aware bijective walk of the tile grid (tile_order.h): what the workgroups resident on one XCD share in its L2.
    // split-K: blockIdx.x = split * nblk + tile; every split accumulates a contiguous range of k-tiles and writes its
    // fp32 partial tile to slab `split` of the scratch buffer (reduced + finished by splitk_finish_kernel)
    const int nblk = ntm * ntn;
    const int split = gbid / nblk;
    const int first_bid = gbid - split * nblk;
    int m0, n0;                          // origin of the output tile being ACCUMULATED (the epilogue's tile)
    {
        int tile_m, tile_n;
        tileorder::tile_of(first_bid, ntm, ntn, order, &tile_m, &tile_n);
        m0 = tile_m * BM;
        n0 = tile_n * BN;
    }

    // ---- per-lane staging coordinates: pass i, this wave writes rows [(i*NW+wave)*RPW, +RPW) of the tile ----
    const int lrow = lane / CPR, lpos = lane % CPR;
    size_t a_off[APASS];                 // ROWS/TEMPORAL: element offset of the row start; CONV2D: image base pixel
    int a_y[APASS], a_x[APASS];          // CONV2D: oy*stride-pad, ox*stride-pad; TEMPORAL: a_y = frame index
    bool a_ok[APASS];
    int a_c8[APASS];                     // source chunk (halfs) after the swizzle
    const int pad = (a.taps == 9) ? 1 : 0;
    size_t b_off[BPASS];
    bool b_ok[BPASS];
    int b_c8[BPASS];
    const int VH = a.Hin * a.up, VW = a.Win * a.up;   // virtual (up-sampled) input grid
    const int ushift = (a.up == 2) ? 1 : 0;
    auto set_stage_tile = [&](const int sm0, const int sn0) {
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int trow = (i * NW + wave) * RPW + lrow;
            a_c8[i] = (lpos ^ chunk_swz<BK>(trow)) * 8;
            const int m = sm0 + trow;
            a_ok[i] = m < a.M;
            const int mm = a_ok[i] ? m : 0;
            if (MODE == IG_ROWS) {
                a_off[i] = (size_t)mm * a.lda;
                a_y[i] = a_x[i] = 0;
            } else if (MODE == IG_CONV2D) {
                const int hw = a.Hout * a.Wout;
                const int n = mm / hw, rem = mm - n * hw;
                const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
                a_off[i] = (size_t)n * a.Hin * a.Win;
                a_y[i] = oy * a.stride - pad;
                a_x[i] = ox * a.stride - pad;
            } else {
                const int fr = mm / a.HW;
                if (a.t_pad) {
                    // frame-sharded clip: A is the padded operand [clip][F + 2][HW][lda] whose frame slots 0 and F + 1 hold the
                    // neighbour ranks' halo frames (zeros at the clip's ends), so no tap is ever masked
                    a_off[i] = ((size_t)mm + (size_t)(2 * (fr / a.F) + 1) * a.HW) * a.lda;
                    a_y[i] = 1;
                } else {
                    a_off[i] = (size_t)mm * a.lda;
                    a_y[i] = fr % a.F;
                }
                a_x[i] = 0;
            }
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            const int trow = (i * NW + wave) * RPW + lrow;
            b_c8[i] = (lpos ^ chunk_swz<BK>(trow)) * 8;
            const int n = sn0 + trow;
            b_ok[i] = (n < a.Nout) && (trow < BN);
            b_off[i] = (size_t)(b_ok[i] ? n : 0) * (a.a_split == 2 ? a.Ktot / 2 : a.Ktot);
        }
    };
    set_stage_tile(m0, n0);

    const half_t* Aptr = (const half_t*)a.A;
    const half_t* Wptr = (const half_t*)a.W;
    const half_t* Rptr = (const half_t*)a.res;

    typedef const void __attribute__((address_space(1)))* gptr_t;
    typedef void __attribute__((address_space(3)))* lptr_t;
    // Paired split-operand walk (a_split == 2): A rows are [hi C | lo C]; k-tile 2p is the hi chunk p of a tap, k-tile 2p+1
    // its lo chunk -- both multiply the SAME weight tile, which is therefore staged once per pair (into the B slot of the
    // pair) and read by both; the weights are the plain [Cout][taps*C] pack.  28 % less global -> LDS traffic than walking
    // [hi | lo] against weights packed twice, which is what limits these tiles.
    const bool paired = (a.a_split == 2);
    const int Cw = a.Cin >> 1;                         // logical channels per tap (paired walk)
    int nk_per_ = (a.Ktot / BK + splitk - 1) / splitk;
    if (paired) nk_per_ = (nk_per_ + 1) & ~1;          // a K split never cuts a (hi, lo) pair
    const int kt_begin = split * nk_per_;

    // asynchronous global -> LDS staging of k-tile kt into buffer buf (no VGPR round trip)
    // (kt = k-tile of the output tile, f = tile count that picks the ring slot: f == kt)
    auto stage = [&](int kt, int f) {
        int k0 = (kt_begin + kt) * BK;                 // K offset of the tile in W rows
        int tap = 0, c0 = k0;
        bool stage_b = true;
        const int buf = f % NSTAGE;
        int bbuf = buf;
        if (paired) {
            const int tpt = a.Cin / BK;                // k-tiles per tap (hi and lo chunks)
            const int ktg = kt_begin + kt;
            tap = ktg / tpt;
            const int r = ktg - tap * tpt, pr = r >> 1, hl = r & 1;
            c0 = pr * BK + hl * Cw;
            k0 = tap * Cw + pr * BK;
            stage_b = (hl == 0);
            bbuf = (f >> 1) % NSTAGE;
        } else if (MODE != IG_ROWS) { tap = k0 / a.Cin; c0 = k0 - tap * a.Cin; }
        int ky = 0, kx = 0;
        if (MODE == IG_CONV2D && a.taps == 9) { ky = tap / 3; kx = tap - 3 * ky; }
        half_t* Ab = As + buf * BM * BK;
        half_t* Bb = Bs + bbuf * BNP * BK;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            bool ok = a_ok[i];
            size_t off;
            if (MODE == IG_ROWS) {
                off = a_off[i] + (paired ? c0 : k0) + a_c8[i];
            } else if (MODE == IG_CONV2D) {
                const int vy = a_y[i] + ky, vx = a_x[i] + kx;
                ok = ok && vy >= 0 && vy < VH && vx >= 0 && vx < VW;
                const int sy = vy >> ushift, sx = vx >> ushift;
                off = (a_off[i] + (size_t)(ok ? sy : 0) * a.Win + (ok ? sx : 0)) * a.lda + c0 + a_c8[i];
            } else {
                const int f = a_y[i] + tap - 1;
                ok = ok && f >= 0 && f < (a.t_pad ? 3 : a.F);
                off = a_off[i] + (ok ? (long)(tap - 1) * a.HW * a.lda : 0) + c0 + a_c8[i];
            }
            const half_t* src = ok ? (Aptr + off) : zeros;          // zero padding = load from a zero page
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Ab + (i * NW + wave) * RPW * BK), 16, 0, 0);
        }
        if (stage_b) {
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                const half_t* src = b_ok[i] ? (Wptr + b_off[i] + k0 + b_c8[i]) : zeros;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Bb + (i * NW + wave) * RPW * BK), 16, 0, 0);
            }
        }
    };

    f4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f4{0.f, 0.f, 0.f, 0.f};

    const int nk_total = a.Ktot / BK;
    const int nk = min(nk_per_, nk_total - kt_begin);
    // NSTAGE-deep LDS ring: D = NSTAGE-1 tiles are in flight; a counted vmcnt (never 0 in steady state) retires only
    // the tile about to be consumed, so the LDS-DMA of later tiles stays in flight across the barrier.
    constexpr int D = NSTAGE - 1;
    constexpr int LPT = APASS + BPASS;       // global_load_lds instructions per lane per tile
    static_assert((D - 1) * LPT <= 63, "vmcnt immediate overflow");
#pragma unroll
    for (int t = 0; t < D; ++t)
        if (t < nk) stage(t, t);

    // fragment read coordinates: row (lane&15) of a 16-row fragment, logical chunk (lane>>4) + 4*kk
    const int frow = lane & 15, fch = lane >> 4;
    int a_rd[MI], b_rd[NI];       // element offset of the fragment row inside a tile buffer
    int a_sw[MI], b_sw[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) { const int r = wm * WM + mi * 16 + frow; a_rd[mi] = r * BK; a_sw[mi] = chunk_swz<BK>(r); }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) { const int r = wn * WN + ni * 16 + frow; b_rd[ni] = r * BK; b_sw[ni] = chunk_swz<BK>(r); }

    // One k-tile = LOAD (every fragment read of the tile issued back to back into distinct registers -- left alone,
    // hipcc re-uses one operand register and serialises {ds_read, lgkmcnt(0), 4 MFMAs} per fragment) + COMPUTE (MFMAs).
    constexpr int KK = BK / 32;
    h8 af[KK][MI], bf[KK][NI];
    auto load_frags = [&](int slot, int bslot) {
        const half_t* Ab = As + slot * BM * BK;
        const half_t* Bb = Bs + bslot * BNP * BK;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bf[kk][ni] = *(const h8*)(Bb + b_rd[ni] + (((kk * 4 + fch) ^ b_sw[ni]) << 3));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[kk][mi] = *(const h8*)(Ab + a_rd[mi] + (((kk * 4 + fch) ^ a_sw[mi]) << 3));
        }
    };
    auto compute = [&]() {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[kk][ni], af[kk][mi], acc[mi][ni], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_16x16x32_f16(af[kk][mi], bf[kk][ni], acc[mi][ni], 0, 0, 0);
        }
    };
    // wait until this wave's LDS-DMA share of tile t has landed, leaving the D-1 younger tiles in flight.  Paired walk:
    // only the even tiles of the window t+1 .. t+D-1 carried a weight tile, so the count of younger loads alternates
    auto wait_tile = [&](int t) {
        if (t + D - 1 >= nk) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
        if (!paired) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * LPT) : "memory"); return; }
        // evens among t+1 .. t+D-1 (local tile indices; the range of a workgroup starts at an even global tile)
        const int nb = ((t + D - 1) >> 1) - (t >> 1);
        if (nb <= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * APASS) : "memory");
        else if (nb == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * APASS + BPASS) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * APASS + 2 * BPASS) : "memory");
    };
    static_assert(D - 1 <= 3, "paired walk: at most two weight tiles among the tiles in flight");
    auto slot_of = [&](int t) { return t % NSTAGE; };
    auto bslot_of = [&](int t) { return paired ? (t >> 1) % NSTAGE : t % NSTAGE; };

    char* const epi_smem = smem_raw;          // the k-loop ring is dead when the epilogue runs
    const int grp = __builtin_amdgcn_readfirstlane(wave) >> 2;                  // the two staggered wave groups (8-wave tiles)
    {
        if constexpr (NW == 8 && (BM / WAVES_M) * (BN / WAVES_N) >= 128 * 64) {
            // (only for 128x64 per-wave tiles: with 64x64 wave tiles the LOAD phase outlasts the MFMAs and staggering loses)
            // Two wave groups (waves 0-3 / 4-7: one wave of each group per SIMD) run half a tile apart: in every barrier
            // interval one group issues its fragment reads + LDS-DMA while the other one issues MFMAs, so the LDS pipe and
            // the matrix pipe are busy at the same time instead of alternating.
            //   interval 2k   : group 0 LOAD(k)      group 1 COMPUTE(k-1)
            //   interval 2k+1 : group 0 COMPUTE(k)   group 1 LOAD(k)
            // Tile k is complete (every wave waited for its own share) before the barrier that opens interval 2k; its ring
            // slot is re-filled from interval 2k+2 on (tile k+NSTAGE-1 is issued during LOAD(k)... of the NEXT tile).
            if (grp == 0) {
                for (int k = 0; k < nk; ++k) {
                    wait_tile(k);
                    __builtin_amdgcn_s_barrier();                     // interval 2k
                    if (k + D < nk) stage(k + D, k + D);
                    load_frags(slot_of(k), bslot_of(k));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();                     // interval 2k+1
                    __builtin_amdgcn_s_setprio(1);
                    compute();
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_barrier();                         // pairs with group 1's last barrier
            } else {
                wait_tile(0);
                __builtin_amdgcn_s_barrier();                         // interval 0 (group 0 loads tile 0)
                for (int k = 0; k < nk; ++k) {
                    __builtin_amdgcn_s_barrier();                     // interval 2k+1
                    if (k + D < nk) stage(k + D, k + D);
                    load_frags(slot_of(k), bslot_of(k));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    if (k + 1 < nk) wait_tile(k + 1);
                    __builtin_amdgcn_s_barrier();                     // interval 2k+2
                    __builtin_amdgcn_s_setprio(1);
                    compute();
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            for (int kt = 0; kt < nk; ++kt) {
                wait_tile(kt);
                __builtin_amdgcn_s_barrier();               // every wave's part of tile kt is visible; slot of tile kt-1 is free
                if (kt + D < nk) stage(kt + D, kt + D);  // streams into LDS under the MFMAs below
                load_frags(slot_of(kt), bslot_of(kt));
                __builtin_amdgcn_sched_barrier(0);
                compute();
                __builtin_amdgcn_sched_barrier(0);
            }
        }

    }

    igemm_epilogue<BM, BN, WAVES_M, WAVES_N, SWAP, MinWaves<BM, BN, NW>::v>(a, acc, m0, n0, wm, wn, lane, wave, split, epi_smem);
}

// =====================================================================================================================
// igemm8_kernel<NI, MODE>: the wide-tile main loop of round 4 -- 256 x (64 NI) x 64 tiles (NI = 4: 256 columns, NI = 5: 320), 8 waves
// as 2 (M) x 4 (N), 128 x 16 NI per wave, in the 8-phase schedule of cdna_hip_programming.md ("The 256^2 8-phase template"):
//
//   * a k-tile (64 deep) lives in LDS as FOUR half-tiles sized by when they are consumed: A0 / A1 = the first / second 64 rows of
//     both wave rows, B0 = fragments 0-1 and B1 = fragments 2.. of all four wave columns (fragment-major: one DMA pass = one
//     fragment of every wave column); two buffers (k-tile parity);
//   * a k-tile is four phases, one C quadrant each: {fragment reads of ONE sub-block (4-12 ds_read_b128), ONE half-tile of LDS-DMA
//     for a later k-tile, barrier, 16-24 MFMAs, barrier}.  The two wave groups (wave row 0 / 1: one wave of each per SIMD) run one
//     barrier apart, so the reads + DMA issue of one group sit under the MFMAs of the other;
//   * the DMA queue is never drained in the loop: half-tiles are issued 5-6 phases before their first read and retired by counted
//     s_waitcnt vmcnt(N) placed ONE phase before that read (the other group's barrier lies in between);
//   * operands come in by buffer_load ... lds: a 32-bit byte offset per staged row (no 64-bit address arithmetic in the loop, the
//     k offset rides in an SGPR), and a masked row -- conv halo, ragged M / N -- is an offset beyond num_records: the hardware
//     writes zeros, no zero page, no select.
//
// Measured against the BK = 32 ring kernel above on the path's shapes (tools/experiments/gemm8_lab.hip, profiles/r04_gemm8_lab_*.txt):
// 1.3-1.5x.  Accumulator fragments keep the layout of igemm_kernel, so the epilogue is shared.
// MODE / a_split / split-K / tile order: as igemm_kernel.  Requirements (checked by the dispatcher): Cin % 64 == 0, swapped
// epilogue, every operand tensor below 2 GiB.
namespace g8 {
constexpr unsigned OOB = 0x80000000u;          // voffset of a masked row: >= num_records of the descriptors below
template <int NI> struct Lds {
    static constexpr int NL = NI - 2;                         // fragments (and DMA passes) of the B1 sub-block
    static constexpr int A0 = 0, A1 = 32 * 1024;              // buffer 1 of a region = its offset ^ the toggle
    static constexpr int B0 = NI == 4 ? 64 * 1024 : 96 * 1024;
    static constexpr int B1 = NI == 4 ? 96 * 1024 : 64 * 1024;
    static constexpr int TOG_A = 16 * 1024, TOG_B0 = 16 * 1024, TOG_B1 = NI == 4 ? 16 * 1024 : 0x30000;      // 64K ^ 0x30000 = 128K
    static constexpr int TOTAL = NI == 4 ? 128 * 1024 : 152 * 1024;
};
// position of a k-tile on the K axis (file scope: a struct local to the kernel, used as a lambda parameter, silently voids the
// kernel's HOST stub -- the handle stays an undefined symbol that only a later executable link reports)
struct KPos { int tap, ac, wk, hl; };
}  // namespace g8

#define G8_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define G8_PHASE_PRE()                                     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_setprio(1);
#define G8_PHASE_POST()                                    \
    __builtin_amdgcn_s_setprio(0);                         \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();                          \
    __builtin_amdgcn_sched_barrier(0);

template <int NI, int MODE>
__global__ __launch_bounds__(512, 2) void igemm8_kernel(IGemmGroup kargs, int per, int ntm, int ntn, int splitk, int order, long red_off) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int gp = __builtin_amdgcn_readfirstlane(blockIdx.x / per);      // grouped launch: see igemm_kernel
    const int gbid = blockIdx.x - gp * per;
    if (gbid >= ntm * ntn * splitk) return;
    const IGemmArgs& a = IGEMM_GROUP_ARGS(gp);
    // in-launch split-K reduction (red_off != -1): fragment slabs at the start of the problem's split-K scratch; ticket words at red_off
    // inside it, or (red_off == -2) the descriptor's own zeroed ticket words
    float* const red_ws = (float*)a.splitk_ws;
    int* const red_cnt = red_off >= 0 ? (int*)((char*)a.splitk_ws + red_off) : (red_off == -2 ? (int*)a.splitk_tickets : nullptr);
    typedef g8::Lds<NI> L;
    constexpr int BM = 256, BN = 64 * NI, BK = 64, WN = 16 * NI, NL = L::NL;
    typedef void __attribute__((address_space(3)))* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int nblk = ntm * ntn;
    const int split = gbid / nblk;
    const int first_bid = gbid - split * nblk;
    int m0, n0;
    {
        int tile_m, tile_n;
        tileorder::tile_of(first_bid, ntm, ntn, order, &tile_m, &tile_n);
        m0 = tile_m * BM;
        n0 = tile_n * BN;
    }

    // ---- k range of this workgroup (split-K: a contiguous range of k-tiles; a (hi, lo) pair of the split-operand walk is never cut) ----
    const bool paired = (a.a_split == 2);
    const int Cw = a.Cin >> 1;
    const int nk_total = a.Ktot / BK;
    int nk_per = (nk_total + splitk - 1) / splitk;
    if (paired) nk_per = (nk_per + 1) & ~1;
    const int kt_begin = split * nk_per;
    const int nk = min(nk_per, nk_total - kt_begin);

    // ---- staging coordinates.  Pass i of a half-tile: this wave writes LDS rows (i*8 + wave)*8 + lane/8, 16-byte chunk lane%8 ----
    const int lrow = lane >> 3, lpos = lane & 7;
    // the chunk swizzle of LDS row j is (j >> 1) & 7 = ((wave & 1) * 4 + lrow / 2) & 7 for every pass: one swizzled source chunk per lane
    const int c8b = ((lpos ^ ((((wave & 1) << 2) + (lrow >> 1)) & 7)) * 8) * 2;        // bytes
    const int pad = (a.taps == 9) ? 1 : 0;
    const int VH = a.Hin * a.up, VW = a.Win * a.up;
    const int ushift = (a.up == 2) ? 1 : 0;
    const int lda2 = (int)a.lda * 2;
    // per staged A row (sub-block sb, pass i = wave row): tile row i*128 + sb*64 + wave*8 + lane/8
    //   ROWS     a_st = byte offset of the row's chunk, or OOB beyond M
    //   TEMPORAL a_st = that offset (a multiple of 16) | bit t set when frame tap t of the row exists (bit 1 = the row itself)
    //   CONV2D   a_st = image << 24 | (oy*stride - pad + 1) << 12 | (ox*stride - pad + 1)   (y field 0xfff beyond M: never in bounds);
    //            a_cur = the offsets of the tap being walked (recomputed when a sub-block's tap changes: once per Cin / 64 k-tiles)
    unsigned a_st[2][2], a_cur[2][2];
    int cur_tap[2] = {-1, -1};
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + i * 128 + sb * 64 + wave * 8 + lrow;
            const bool ok = m < a.M;
            const int mm = ok ? m : 0;
            a_cur[sb][i] = g8::OOB;
            if (MODE == IG_ROWS) {
                a_st[sb][i] = ok ? (unsigned)(mm * lda2 + c8b) : g8::OOB;
            } else if (MODE == IG_CONV2D) {
                const int hw = a.Hout * a.Wout;
                const int n = mm / hw, rem = mm - n * hw;
                const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
                const int yf = ok ? oy * a.stride - pad + 1 : 0xfff;
                a_st[sb][i] = ((unsigned)n << 24) | ((unsigned)yf << 12) | (unsigned)(ox * a.stride - pad + 1);
            } else {
                const int fr = mm / a.HW;
                unsigned base, bits;
                if (a.t_pad) {
                    // frame-sharded clip: A is the padded operand [clip][F + 2][HW][lda]; slots 0 / F + 1 hold the halo frames
                    base = (unsigned)((mm + (2 * (fr / a.F) + 1) * a.HW) * lda2 + c8b);
                    bits = 7u;
                } else {
                    const int f = fr % a.F;
                    base = (unsigned)(mm * lda2 + c8b);
                    bits = 2u | (f >= 1 ? 1u : 0u) | (f + 1 < a.F ? 4u : 0u);
                }
                a_st[sb][i] = ok ? (base | bits) : 0u;
            }
        }
    // weight rows: LDS row f*64 + wn*16 + r of a B sub-block = fragment f (= DMA pass f) of wave column wn, so a lane stages the
    // SAME row of consecutive fragments in consecutive passes: one offset per sub-block, + f*16 rows through the scalar offset
    // (Nout is a multiple of the tile width here: no masked weight rows)
    const int ldw2 = (paired ? a.Ktot / 2 : a.Ktot) * 2;      // bytes per weight row
    const unsigned b0_vo = (unsigned)((n0 + ((wave * 8 + lrow) >> 4) * WN + ((wave * 8 + lrow) & 15)) * ldw2 + c8b);
    const unsigned b1_vo = b0_vo + 32u * (unsigned)ldw2;
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.A), 0, (int)0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W), 0, (int)0x80000000u, 0x00020000);

    // position of a k-tile in the K axis: tap, channel offset of the A chunk (bytes), k offset in a weight row (bytes).  The stream
    // is staged in order, so positions advance incrementally (no division per k-tile)
    typedef g8::KPos KPos;
    auto kpos_init = [&](int ktg) __attribute__((always_inline)) {
        KPos p;
        if (paired) {
            const int tpt = a.Cin / BK;
            p.tap = ktg / tpt;
            const int r = ktg - p.tap * tpt, pr = r >> 1;
            p.hl = r & 1;
            p.ac = (pr * BK + p.hl * Cw) * 2;
            p.wk = (p.tap * Cw + pr * BK) * 2;
        } else {
            const int k0 = ktg * BK;
            p.tap = (MODE == IG_ROWS) ? 0 : k0 / a.Cin;
            p.ac = (k0 - p.tap * a.Cin) * 2;
            p.wk = k0 * 2;
            p.hl = 0;
        }
        return p;
    };
    auto kpos_next = [&](KPos p) __attribute__((always_inline)) {
        if (paired) {
            if (p.hl == 0) { p.hl = 1; p.ac += Cw * 2; }
            else {
                p.hl = 0; p.ac += (BK - Cw) * 2; p.wk += BK * 2;
                if (p.ac == Cw * 2) { p.ac = 0; ++p.tap; }
            }
        } else {
            p.ac += BK * 2; p.wk += BK * 2;
            if (MODE != IG_ROWS && p.ac == a.Cin * 2) { p.ac = 0; ++p.tap; }
        }
        return p;
    };

    // LDS is addressed by byte offset (the kernel declares no static LDS, so the dynamic segment starts at 0): fragment reads are
    // `per-lane base + immediate`, DMA destinations scalar arithmetic -- no relocation adds in the loop
    typedef const h8 __attribute__((address_space(3)))* lds_h8_t;
#define G8_LDS_DST(off) ((lptr_t)(size_t)(unsigned)(off))        // (a lambda returning an LDS pointer silently voids the host stub)
    // A half-tile sb of the k-tile at p -> buffer buf
    auto stage_a = [&](const int sb, const KPos p, const int buf) __attribute__((always_inline)) {
        const int dst = ((sb ? L::A1 : L::A0) ^ (buf ? L::TOG_A : 0)) + wave * 1024;
        if (MODE == IG_CONV2D && p.tap != cur_tap[sb]) {
            cur_tap[sb] = p.tap;
            int ky = 0, kx = 0;
            if (a.taps == 9) { ky = (p.tap * 11) >> 5; kx = p.tap - 3 * ky; }
            const int hwin = a.Hin * a.Win;
            // (the operands are made opaque here: hoisted out of the k-loop, the unpacked fields of a_st cost eight more registers
            // than the loop has, and their spill reloads drain the DMA queue)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int c8u = (((ln & 7) ^ ((((wave & 1) << 2) + (ln >> 4)) & 7)) * 8) * 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                unsigned st = a_st[sb][i];
                asm volatile("" : "+v"(st));
                const int vy = (int)((st >> 12) & 0xfff) - 1 + ky, vx = (int)(st & 0xfff) - 1 + kx;
                const bool ok = (unsigned)vy < (unsigned)VH && (unsigned)vx < (unsigned)VW;
                const int pix = (int)(st >> 24) * hwin + (vy >> ushift) * a.Win + (vx >> ushift);
                // (multiply and add kept apart: fused into v_mad_u64_u32 they need a register PAIR, and the 256-register 320-wide tile
                // then spills one register around it -- a scratch reload inside the k-loop, which the in-order vmcnt turns into a drain
                // of the DMA queue; tests/test_kernel_resources.py)
                unsigned poff = (unsigned)pix * (unsigned)lda2;
                asm volatile("" : "+v"(poff));
                a_cur[sb][i] = ok ? poff + (unsigned)c8u : g8::OOB;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned vo;
            if (MODE == IG_ROWS) {
                vo = a_st[sb][i];
            } else if (MODE == IG_CONV2D) {
                vo = a_cur[sb][i];
            } else {
                const unsigned st = a_st[sb][i];
                vo = ((st >> p.tap) & 1u) ? (st & ~15u) + (unsigned)((p.tap - 1) * a.HW * lda2) : g8::OOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, G8_LDS_DST(dst + i * 8192), 16, vo, p.ac, 0, 0);
        }
    };
    auto stage_b0 = [&](const KPos p, const int buf) __attribute__((always_inline)) {
        const int dst = (L::B0 ^ (buf ? L::TOG_B0 : 0)) + wave * 1024;
#pragma unroll
        for (int f = 0; f < 2; ++f) __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, G8_LDS_DST(dst + f * 8192), 16, b0_vo, p.wk + f * 16 * ldw2, 0, 0);
    };
    auto stage_b1 = [&](const KPos p, const int buf) __attribute__((always_inline)) {
        const int dst = (L::B1 ^ (buf ? L::TOG_B1 : 0)) + wave * 1024;
#pragma unroll
        for (int f = 0; f < NL; ++f) __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, G8_LDS_DST(dst + f * 8192), 16, b1_vo, p.wk + f * 16 * ldw2, 0, 0);
    };

    // ---- fragment reads: one per-lane base per operand and k-step (second k-step = logical chunk + 4 = address ^ 64), everything
    //      else an immediate; the buffer toggles by XOR ----
    const int frow = lane & 15, fch = lane >> 4;
    const int sw = (frow >> 1) & 7;
    int ra0 = (wm * 64 + frow) * 128 + ((fch ^ sw) << 4);                           // + A0 | A1 + mi*2048
    int rb00 = L::B0 + (wn * 16 + frow) * 128 + ((fch ^ sw) << 4);                  // + f*8192
    int rb10 = L::B1 + (wn * 16 + frow) * 128 + ((fch ^ sw) << 4);                  // + f*8192
    int ra1 = ra0 ^ 64, rb01 = rb00 ^ 64, rb11 = rb10 ^ 64;

    f4 acc[8][NI];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f4{0.f, 0.f, 0.f, 0.f};

    h8 af[2][4], b0f[2][2], b1f[2][NL];       // [k-step][fragment]
    auto read_a = [&](const int sb) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[kk][mi] = *(lds_h8_t)(size_t)(unsigned)((kk ? ra1 : ra0) + (sb ? L::A1 : L::A0) + mi * 2048);
    };
    auto read_b0 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int f = 0; f < 2; ++f) b0f[kk][f] = *(lds_h8_t)(size_t)(unsigned)((kk ? rb01 : rb00) + f * 8192);
    };
    auto read_b1 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int f = 0; f < NL; ++f) b1f[kk][f] = *(lds_h8_t)(size_t)(unsigned)((kk ? rb11 : rb10) + f * 8192);
    };
    // one C quadrant: rows as*64.. of the wave tile x fragments of one B sub-block (operands swapped: D = W . A^T)
    auto mma_b0 = [&](const int as) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int f = 0; f < 2; ++f)
                    acc[as * 4 + mi][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0f[kk][f], af[kk][mi], acc[as * 4 + mi][f], 0, 0, 0);
    };
    auto mma_b1 = [&](const int as) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int f = 0; f < NL; ++f)
                    acc[as * 4 + mi][2 + f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1f[kk][f], af[kk][mi], acc[as * 4 + mi][2 + f], 0, 0, 0);
    };

    // Issue order of the half-tiles: A0, B0, B1, A1 of k-tile 0, A0, B0 of k-tile 1 (prologue), then one per phase --
    //   phase 0 of k-tile e: B1[e+1]   phase 1: A1[e+1]   phase 2: A0[e+2]   phase 3: B0[e+2]
    // -- each into a region whose last read lies at least two phases back (the other wave group's reads of phase p are only known
    // complete after ITS second barrier of phase p).  Reads: A0[e], B0[e] in phase 0, B1[e] in phase 1, A1[e] in phase 2; b0 stays in
    // registers for phase 3.  A half-tile is waited for ONE phase before its first read: B1[e] in phase 0, A1[e] in phase 1,
    // A0 / B0[e+1] in phase 3.  The counter retires in order, so "at most n younger loads in flight" = the wanted one has landed;
    // the younger ones are always four half-tiles = 6 + NL loads while the stream lasts (n1: k-tile e+1 exists, n2: e+2 exists).
    // (the immediates are spelled out per NL: an "n" operand that depends on a template parameter silently voids the HOST stub of
    // the kernel -- the handle stays an undefined symbol and only the tools' link step notices)
#define G8_VMCNT_W4() do { if constexpr (NL == 2) G8_VMCNT(8); else G8_VMCNT(9); } while (0)     /* four half-tiles: 6 + NL loads */
#define G8_VMCNT_W2() do { if constexpr (NL == 2) G8_VMCNT(4); else G8_VMCNT(5); } while (0)     /* B1 + A1: NL + 2 loads */
    KPos p1, p2;                              // positions of k-tiles e+1 and e+2
    {
        const KPos p0 = kpos_init(kt_begin);
        stage_a(0, p0, 0); stage_b0(p0, 0); stage_b1(p0, 0); stage_a(1, p0, 0);
        p1 = kpos_next(p0);
        if (nk > 1) { stage_a(0, p1, 1); stage_b0(p1, 1); G8_VMCNT_W4(); } else G8_VMCNT_W2();
        p2 = kpos_next(p1);
    }
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();          // the second wave group runs one barrier behind

    for (int e = 0; e < nk; ++e) {
        const bool n1 = e + 1 < nk, n2 = e + 2 < nk;
        const int nb = (e + 1) & 1;                     // buffer of k-tile e+1; k-tile e+2 goes where k-tile e is
        // phase 0: a0, b0 | stage B1[e+1] | wait B1[e]
        read_b0();
        __builtin_amdgcn_sched_barrier(0);
        read_a(0);
        if (n1) { stage_b1(p1, nb); G8_VMCNT_W4(); } else G8_VMCNT(2);
        G8_PHASE_PRE();
        mma_b0(0);
        G8_PHASE_POST();
        // phase 1: b1 | stage A1[e+1] | wait A1[e]
        read_b1();
        if (n1) { stage_a(1, p1, nb); G8_VMCNT_W4(); } else G8_VMCNT(0);
        G8_PHASE_PRE();
        mma_b1(0);
        G8_PHASE_POST();
        // phase 2: a1 | stage A0[e+2]
        read_a(1);
        if (n2) stage_a(0, p2, nb ^ 1);
        G8_PHASE_PRE();
        mma_b1(1);
        G8_PHASE_POST();
        // phase 3: (b0 still in registers) | stage B0[e+2] | wait A0[e+1], B0[e+1]
        if (n2) { stage_b0(p2, nb ^ 1); G8_VMCNT_W4(); } else if (n1) G8_VMCNT_W2();
        G8_PHASE_PRE();
        mma_b0(1);
        G8_PHASE_POST();
        ra0 ^= L::TOG_A; ra1 ^= L::TOG_A; rb00 ^= L::TOG_B0; rb01 ^= L::TOG_B0; rb10 ^= L::TOG_B1; rb11 ^= L::TOG_B1;
        p1 = p2;
        p2 = kpos_next(p2);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();

    int slab = split;
    if (red_cnt != nullptr) {
        // In-launch split-K reduction (cdna_hip_programming.md "in-launch split-K reduction", MI355X_MICROARCH.md
        // "Workgroup dispatch ... inter-workgroup visibility"): every split writes its accumulators to its slab of the tile in
        // FRAGMENT layout (16 bytes per lane, 8 KiB contiguous per instruction), publishes them (write-through stores, drained) and takes a
        // ticket; the workgroup that draws the last ticket acquires once, adds the slabs in split order -- a fixed order whoever arrives
        // last, so the result is bit-reproducible -- and goes on.  No workgroup waits for another: the others just exit.
        //   2..4 splits: one level -- the last arriver of the tile runs the full epilogue (round 4).
        //   5..16 splits (round 6, opt-in: CTRL_SPLITK_INLAUNCH=all): two levels -- the splits form groups of four; the last arriver of a
        //   GROUP sums the group's slabs into the group's first slab, publishes it and takes the tile's second-level ticket; the last of
        //   those sums the <= 4 group sums and runs the epilogue.  Nobody reads more than four slabs -- and it still LOSES to the finish
        //   kernel (see launch8): a lone workgroup streams its slabs at a fraction of what the finish kernel's 4096 workgroups get.
        // It replaces the fp32 row slabs + splitk_finish_kernel launch (round 3: 36 launches, 1.1 ms per step).
        const int tile = first_bid;
        constexpr int NF = 8 * NI;
        f4* const tile_ws = (f4*)red_ws + (size_t)tile * splitk * NF * 512;
        const bool two = splitk > 4;
        const int gsz = two ? 4 : splitk;                       // splits per group
        const int ngrp = (splitk + gsz - 1) / gsz;
        const int grp_i = split / gsz;
        const int g_first = grp_i * gsz, g_n = min(gsz, splitk - g_first);
        // ticket words of a tile: [0] = the last level, [1 + g] = group g (two levels only)
        int* const cnt_last = red_cnt + (size_t)tile * (two ? 1 + 4 : 1);
        int* const flag = (int*)smem_raw;
        // The slab leaves by WRITE-THROUGH stores (sc1) and is published by draining them: a release fence writes back every
        // dirty line of the XCD's L2 -- with 320 KB freshly written by each of the XCD's 32 workgroups that cost 40-66 us per launch
        // (MI355X_MICROARCH.md price list, "publish-large"), more than the finish kernel it replaces.
        auto publish = [&](const int slab_i, int* const counter, const int last_ticket) __attribute__((always_inline)) {
            const __amdgpu_buffer_rsrc_t ws_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(tile_ws + (size_t)slab_i * NF * 512), 0, NF * 512 * 16, 0x00020000);
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    u4 v;
                    __builtin_memcpy(&v, &acc[mi][ni], 16);
                    __builtin_amdgcn_raw_buffer_store_b128(v, ws_rs, ((mi * NI + ni) * 512 + tid) * 16, 0, /*sc1*/ 16);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                       // (also: every wave is done with the LDS ring / the flag word)
            if (tid == 0) {
                const int ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ticket == last_ticket) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                *flag = ticket;
            }
            __syncthreads();
            const bool last = *flag == last_ticket;
            __syncthreads();                                       // the flag word is re-used by the next level / the epilogue's staging area
            return last;
        };
        auto sum_slabs = [&](const int first, const int count, const int stride) __attribute__((always_inline)) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f4{0.f, 0.f, 0.f, 0.f};
            for (int sp = 0; sp < count; ++sp) {
                const f4* src = tile_ws + (size_t)(first + sp * stride) * NF * 512 + tid;
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] += src[(mi * NI + ni) * 512];
            }
        };
        if (!publish(split, two ? cnt_last + 1 + grp_i : cnt_last, g_n - 1)) return;
        sum_slabs(g_first, g_n, 1);
        if (two) {
            if (!publish(g_first, cnt_last, ngrp - 1)) return;      // the group's sum replaces the group's first slab (every slab of the group has been read)
            sum_slabs(0, ngrp, gsz);
        }
        slab = 0;
    }
    igemm_epilogue<BM, BN, 2, 4, true, 1>(a, acc, m0, n0, wm, wn, lane, wave, slab, smem_raw);
}

// split-K tail: out = epilogue( sum_s slab[s] ), 8 consecutive columns per thread (row-major outputs only)
__global__ __launch_bounds__(256) void splitk_finish_kernel(IGemmArgs a, const float* __restrict__ ws, int splitk) {
    const int nc8 = a.Nout >> 3;
    const size_t total = (size_t)a.M * nc8, slab = (size_t)a.M * a.Nout;
    const IGemmSeg sg = a.seg[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / nc8;
        const int col = (int)(i - row * nc8) * 8;
        float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < splitk; ++s) {
            const float* p = ws + s * slab + row * a.Nout + col;
            const f4 u = *(const f4*)p, v = *(const f4*)(p + 4);
            x[0] += u[0]; x[1] += u[1]; x[2] += u[2]; x[3] += u[3]; x[4] += v[0]; x[5] += v[1]; x[6] += v[2]; x[7] += v[3];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (a.bias) x[j] += a.bias[col + j];
            if (a.rowvec) x[j] += a.rowvec[(row / a.rows_per_img) * a.rowvec_ld + col + j];
            if (a.act == 1) x[j] = silu_f(x[j]);
        }
        if (a.res) {
            if (a.res_f32) {
                const float* rp = (const float*)a.res + res_row_of(a, (int)row) * a.ldres + col;
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] += rp[j];
            } else {
                const h8 rr = *(const h8*)((const half_t*)a.res + res_row_of(a, (int)row) * a.ldres + col);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] += (float)rr[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] *= a.scale;
        if (a.blend_mix) {
            const float al = __builtin_amdgcn_rcpf(1.0f + __expf(-a.blend_mix[0]));
            float bx[8];
            if (a.blend_f32) {
                const float* bp = (const float*)a.blend_x + row * a.ld_blend + col;
#pragma unroll
                for (int j = 0; j < 8; ++j) bx[j] = bp[j];
            } else {
                const h8 bb = *(const h8*)((const half_t*)a.blend_x + row * a.ld_blend + col);
#pragma unroll
                for (int j = 0; j < 8; ++j) bx[j] = (float)bb[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = al * bx[j] + (1.0f - al) * x[j];
        }
        const size_t o = row * sg.ld + col;
        if (sg.dtype == DT_F32) {
            *(f4*)((float*)sg.out + o) = f4{x[0], x[1], x[2], x[3]};
            *(f4*)((float*)sg.out + o + 4) = f4{x[4], x[5], x[6], x[7]};
        } else if (sg.dtype == DT_F16) {
            h8 pk;
#pragma unroll
            for (int j = 0; j < 8; ++j) pk[j] = (half_t)x[j];
            *(h8*)((half_t*)sg.out + o) = pk;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) ((u16*)sg.out)[o + j] = f32_to_bf16(x[j]);
        }
        if (a.nonfinite && (a.out16 != nullptr || sg.dtype == DT_F16)) {      // range check (ctrl_igemm_desc::nonfinite), as the GEMM epilogues
            bool bad = false;
#pragma unroll
            for (int j = 0; j < 8; ++j) bad = bad || out_of_half(x[j]);
            if (bad) *(volatile int32_t*)a.nonfinite = 1;
        }
        if (a.out16) {
            h8 pk;
#pragma unroll
            for (int j = 0; j < 8; ++j) pk[j] = (half_t)x[j];
            *(h8*)((half_t*)a.out16 + row * a.ld16 + col) = pk;
            if (a.out16_lo_off) {
                h8 lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) lo[j] = (half_t)(x[j] - (float)pk[j]);
                *(h8*)((half_t*)a.out16 + row * a.ld16 + a.out16_lo_off + col) = lo;
            }
        }
    }
}

const half_t* zero_page() { return (const half_t*)device_zero_page(); }

// ---- tile walk order (tile_order.h) ----
// CTRL_IGEMM_ORDER / ctrl_igemm_set_order(): "auto" (default, below), "legacy", or a forced "m,G" / "n,G".
struct OrderSpec { int kind; int mode; int group; };       // kind 0 legacy, 1 auto, 2 forced
OrderSpec g_order_spec = {-1, 0, 0};

OrderSpec parse_order(const char* t) {
    if (!t || !*t || !strcmp(t, "legacy") || !strcmp(t, "0")) return {0, 0, 0};
    if (!strcmp(t, "auto")) return {1, 0, 0};
    if ((t[0] == 'm' || t[0] == 'n') && t[1] == ',') {
        const int g = atoi(t + 2);
        if (g >= 0 && g < 65536) return {2, t[0] == 'm' ? tileorder::ORDER_XCD_M : tileorder::ORDER_XCD_N, g};
    }
    return {-2, 0, 0};
}

// "auto" = what was measured (tools/gemm_order_bench.cpp -> profiles/r02_tile_order_experiment.txt): a token GEMM whose
// weights do not fit in an XCD's L2 beside the activation / output streams (the 512 -> 4096 GEGLU projection, 4 MiB) walks
// its weight panels in groups of <= 1 MiB under the legacy XCD split of the rows: 0.844 -> 0.786 ms at M = 131072, results
// bit-identical; the same walk is neutral on the other wide GEMMs of the path (M = 32768 and the ControlNet's GEGLUs), and
// splitting the weight panels over the XCDs instead ("n,G") never won.  Everything else keeps the legacy walk.
int plan_order(const IGemmArgs& a, int BM, int BN, int ntm, int ntn) {
    if (g_order_spec.kind == -1) {
        const char* e = policy_raw(P_IGEMM_ORDER);
        g_order_spec = e ? parse_order(e) : OrderSpec{1, 0, 0};
        if (g_order_spec.kind == -2) { fprintf(stderr, "ctrl: CTRL_IGEMM_ORDER not understood, using auto\n"); g_order_spec = {1, 0, 0}; }
    }
    (void)BM;
    const OrderSpec sp = g_order_spec;
    if (sp.kind == 0) return 0;
    if (sp.kind == 2) return tileorder::make_order(sp.mode, sp.group);
    if (a.mode != IG_ROWS || (ntm & 7) != 0) return 0;
    const double panel = (double)BN * a.Ktot * 2.0, weights = (double)a.Nout * a.Ktot * 2.0;      // bytes
    if (weights <= 2.0 * 1024 * 1024) return 0;
    int G = (int)(1024.0 * 1024.0 / panel);
    if (G < 1) G = 1;
    return G < ntn ? tileorder::make_order(tileorder::ORDER_XCD_M, G) : 0;
}

// descriptors of the grouped launch being dispatched (op_igemm_group): [0] is the problem the dispatcher sees
thread_local const IGemmArgs* t_grp = nullptr;
thread_local int t_grp_n = 1;
inline int grp_count() { return t_grp ? t_grp_n : 1; }
inline const IGemmArgs& grp_at(const IGemmArgs& a, int i) { return t_grp ? t_grp[i] : a; }
// workgroups per problem of a launch: a group pads them to a multiple of 8 (XCD of a workgroup = its local index % 8, tile_order.h)
inline int grp_per(int nblk) { return grp_count() > 1 ? ((nblk + 7) & ~7) : nblk; }
// split-K in the finish-kernel form: the GEMM writes fp32 row slabs, every epilogue term is applied once by the finish kernel
inline IGemmArgs splitk_slab_desc(const IGemmArgs& a) {
    IGemmArgs p = a;
    p.bias = nullptr; p.rowvec = nullptr; p.res = nullptr; p.res_f32 = 0; p.act = 0; p.scale = 1.f; p.out16 = nullptr;
    p.blend_mix = nullptr; p.blend_x = nullptr;
    p.nseg = 1;
    p.seg[0] = IGemmSeg{a.splitk_ws, a.Nout, 0, a.Nout, SEG_ROW, DT_F32, 1, 0};
    return p;
}
int launch_splitk_finish(const IGemmArgs& a, int splitk, hipStream_t s) {
    for (int i = 0; i < grp_count(); ++i) {
        const IGemmArgs& ai = grp_at(a, i);
        const size_t total = (size_t)ai.M * (ai.Nout / 8);
        size_t blocks = (total + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        PROF_WORK(0, 4.0 * splitk * ai.M * ai.Nout);
        prof_detail("M%d N%d splitk%d", ai.M, ai.Nout, splitk);
        LAUNCH("splitk_finish", splitk_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ai, (const float*)ai.splitk_ws, splitk);
    }
    return 0;
}
void prof_igemm(const IGemmArgs& a, const char* extra) {
    const int G = grp_count();
    // algorithmic work = the reference op's: a split operand doubles the K the kernel walks, not the FLOPs that count
    const double kalg = a.a_split ? 0.5 * a.Ktot : (double)a.Ktot;
    PROF_WORK(G * 2.0 * a.M * a.Nout * kalg, G * 2.0 * ((double)a.M * a.Cin + (double)a.Nout * a.Ktot + (double)a.M * a.Nout));
    char grp[16] = "";
    if (G > 1) snprintf(grp, sizeof(grp), " x%d", G);
    prof_detail("M%d N%d K%d taps%d%s%s%s%s", a.M, a.Nout, a.Ktot, a.taps, a.a_split ? " split-A" : "", a.geglu ? " geglu" : "", extra, grp);
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int NSTAGE, int MODE, bool SWAP>
int launch_cfg2(const IGemmArgs& a, hipStream_t s, int splitk = 1) {
    constexpr int PASSROWS = WAVES_M * WAVES_N * (64 / (BK / 8));
    constexpr int BNP = (BN + PASSROWS - 1) / PASSROWS * PASSROWS;
    constexpr size_t ring = (size_t)NSTAGE * (BM + BNP) * BK * sizeof(half_t);
    constexpr int stage_w = (BN / WAVES_N) > (BM / WAVES_M) ? (BN / WAVES_N) : (BM / WAVES_M);     // row / transposed staging
    constexpr size_t stage_bytes = (size_t)WAVES_M * WAVES_N * 16 * (stage_w + 4) * sizeof(float);
    constexpr size_t smem = ring > stage_bytes ? ring : stage_bytes;
    static bool attr_done[kMaxDevices] = {};       // function attributes are per device
    const int dev = cur_device();
    if (!attr_done[dev]) {
        HIP_TRY(hipFuncSetAttribute((const void*)igemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, NSTAGE, MODE, SWAP>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_done[dev] = true;
    }
    const half_t* zeros = zero_page();
    CTRL_CHECK(zeros != nullptr, "igemm: could not allocate the zero page");
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.Nout + BN - 1) / BN;
    const int order = plan_order(a, BM, BN, ntm, ntn);
    const int G = grp_count();
    const int per = grp_per(ntm * ntn * splitk);
    const char* tag = MODE == IG_ROWS ? "igemm_rows" : (MODE == IG_CONV2D ? "igemm_conv" : "igemm_temporal");
    const auto sym = [&]() { prof_symbol("igemm_kernel<%d, %d, %d, %d, %d, %d, %d, %s>", BM, BN, BK, WAVES_M, WAVES_N, NSTAGE, MODE, SWAP ? "true" : "false"); };
    IGemmGroup grp;
    if (splitk > 1) {
        // partial sums go to fp32 slabs [splitk][M][Nout]; every epilogue term is applied once, by the finish kernel
        for (int i = 0; i < G; ++i) grp.a[i] = splitk_slab_desc(grp_at(a, i));
        char ex[32]; snprintf(ex, sizeof(ex), " splitk%d", splitk);
        prof_igemm(a, ex);
        sym();
        LAUNCH(tag, (igemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, NSTAGE, MODE, SWAP>), dim3(per * G),
               dim3(WAVES_M * WAVES_N * 64), smem, s, grp, per, ntm, ntn, zeros, splitk, order);
        return launch_splitk_finish(a, splitk, s);
    }
    for (int i = 0; i < G; ++i) grp.a[i] = grp_at(a, i);
    prof_igemm(a, "");
    sym();
    LAUNCH(tag, (igemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, NSTAGE, MODE, SWAP>), dim3(per * G), dim3(WAVES_M * WAVES_N * 64), smem, s,
           grp, per, ntm, ntn, zeros, 1, order);
    return 0;
}

// Aligned row-major outputs and single aligned transposed (NCHW / V^T) outputs use the swapped-operand kernel (LDS-staged
// vector epilogue); mixed or unaligned segment lists keep the natural orientation and its scalar epilogue.
bool can_swap(const IGemmArgs& a) {
    bool swap = (a.Nout % 16 == 0);
    // a transposed segment must be the last one (LDS-transposed epilogue, 8-token pieces: an image's tokens start 16-byte aligned); in
    // front of it only row-major segments, and then it starts at a multiple of 64 columns (mixed_boundary)
    const int nrow = (a.seg[a.nseg - 1].fmt == SEG_TRANSPOSED) ? a.nseg - 1 : a.nseg;
    if (nrow < a.nseg) {
        const IGemmSeg& g = a.seg[a.nseg - 1];
        swap = swap && !a.geglu && !a.out16 && !a.blend_mix && g.L > 0 && (g.L % 8 == 0) && (g.ld % 8 == 0) && (a.M % 8 == 0) &&
               (((uintptr_t)g.out & 15) == 0) && (nrow == 0 || (g.col_begin % 64 == 0 && !a.res && !a.rowvec));
    }
    for (int i = 0; i < nrow; ++i)
        swap = swap && a.seg[i].fmt == SEG_ROW && (a.seg[i].ld % 8 == 0) && (((uintptr_t)a.seg[i].out & 15) == 0) &&
               (a.seg[i].col_begin % 8 == 0);
    if (a.res) swap = swap && (a.ldres % 8 == 0) && (((uintptr_t)a.res & 15) == 0);
    if (a.out16) swap = swap && (a.ld16 % 8 == 0) && (((uintptr_t)a.out16 & 15) == 0) && (a.out16_lo_off % 8 == 0);
    if (a.blend_mix) swap = swap && a.blend_x && (a.ld_blend % 8 == 0) && (((uintptr_t)a.blend_x & 15) == 0);
    if (a.bias) swap = swap && (((uintptr_t)a.bias & 15) == 0);
    if (a.rowvec) swap = swap && (a.rowvec_ld % 4 == 0) && (((uintptr_t)a.rowvec & 15) == 0);
    return swap;
}

// the 8-phase wide-tile kernel (igemm8_kernel) can take this problem: swapped epilogue, whole 64-deep k-tiles per tap, 32-bit offsets
// CTRL_IGEMM8: "0" never, "force" whenever the problem is eligible (any grid size: the parity tests run their small shapes
// through it this way), default = where the grid fills the chip
int g_igemm8_mode = -1;
int igemm8_mode() {
    if (g_igemm8_mode >= 0) return g_igemm8_mode;      // ctrl_igemm_set_wide (test ABI)
    const char* e = policy_raw(P_IGEMM8);
    return !e ? 1 : (e[0] == '0' ? 0 : (!strcmp(e, "force") ? 2 : 1));
}
bool can_use8(const IGemmArgs& a) {
    if (!igemm8_mode() || !can_swap(a)) return false;
    if (a.Cin % 64 != 0 || (a.a_split == 2 && ((a.Cin / 2) % 64 != 0 || a.mode == IG_TEMPORAL))) return false;
    // conv rows are packed as image (8 bits) | y + 1 (12 bits) | x + 1 (12 bits)
    if (a.mode == IG_CONV2D && ((double)a.M / ((double)a.Hout * a.Wout) > 256.0 || a.Hin * a.up > 4000 || a.Win * a.up > 4000 ||
                                 a.Hout * a.stride > 4000 || a.Wout * a.stride > 4000)) return false;
    double abytes;
    if (a.mode == IG_CONV2D) abytes = (double)a.M / ((double)a.Hout * a.Wout) * a.Hin * a.Win * a.lda * 2.0;
    else if (a.mode == IG_TEMPORAL && a.t_pad) abytes = ((double)a.M + 2.0 * a.HW * ((double)a.M / ((double)a.F * a.HW))) * a.lda * 2.0;
    else abytes = (double)a.M * a.lda * 2.0;
    return abytes < 2147483648.0 && (double)a.Nout * a.Ktot * 2.0 < 2147483648.0;
}

template <int NI, int MODE>
int launch8(const IGemmArgs& a, hipStream_t s, int splitk = 1) {
    constexpr int BM = 256, BN = 64 * NI;
    constexpr size_t ring = g8::Lds<NI>::TOTAL;
    constexpr int stage_w = 128 > 16 * NI ? 128 : 16 * NI;                                            // row / transposed staging of the shared epilogue
    constexpr size_t stage_bytes = (size_t)8 * 16 * (stage_w + 4) * sizeof(float);
    constexpr size_t smem = ring > stage_bytes ? ring : stage_bytes;
    static bool attr_done[kMaxDevices] = {};
    const int dev = cur_device();
    if (!attr_done[dev]) {
        HIP_TRY(hipFuncSetAttribute((const void*)igemm8_kernel<NI, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_done[dev] = true;
    }
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.Nout + BN - 1) / BN;
    // Tile walk: the legacy one (every XCD a contiguous range of the M-major list) unless an order is forced.  The grouped
    // weight-panel walk that pays for the BK = 32 kernels re-fetches every activation panel once per group, and at this kernel's
    // rate that traffic is the limiter: M131072 N2048 K2048 1.52 ms grouped vs 1.07 ms legacy (profiles/r04_gemm_tile_order_8phase.txt)
    (void)plan_order(a, BM, BN, ntm, ntn);          // (parses CTRL_IGEMM_ORDER on first use)
    const int order = g_order_spec.kind == 2 ? tileorder::make_order(g_order_spec.mode, g_order_spec.group) : 0;
    const int G = grp_count();
    const int per = grp_per(ntm * ntn * splitk);
    const char* tag = MODE == IG_ROWS ? "igemm_rows" : (MODE == IG_CONV2D ? "igemm_conv" : "igemm_temporal");
    const auto sym = [&]() { prof_symbol("igemm8_kernel<%d, %d>", NI, MODE); };
    IGemmGroup grp;
    if (splitk > 1) {
        // 2..4 splits: reduced inside the launch by the last-arriving workgroup of every tile (see the kernel); the workspace then
        // holds tile-shaped fragment slabs [tile][split][fragment][thread] + one ticket word per tile (zeroed by a memset node)
        const size_t slabs = (size_t)ntm * ntn * splitk * BM * BN * sizeof(float);
        const bool own_tickets = a.splitk_tickets != nullptr;       // (igemm_same_form: all of the group or none)
        const size_t twords = (size_t)ntm * ntn * (splitk > 4 ? 5 : 1);      // ticket words: see the kernel
        const size_t need = own_tickets ? slabs : (slabs + 255) / 256 * 256 + twords * sizeof(int);
        // CTRL_SPLITK_INLAUNCH: 0 = always the finish kernel; default = in-launch for 2..4 splits; "all" = also the two-level form for 5..16
        // splits -- built and measured in round 6, NOT the default: the last arrivers read their <= 4 slabs of 256-320 KB alone, at the
        // 60-120 GB/s one workgroup gets (MI355X_MICROARCH.md "handoff-payload"), twice; M512 N1280 K11520 x15: 52 + 14 us (GEMM + finish
        // kernel over the whole chip) -> 128 us, the ControlNet 8.16 -> 9.14 ms (profiles/r06_splitk_two_level.txt)
        const char* const il = policy_raw(P_SPLITK_INLAUNCH);
        const int inlaunch = (il && il[0] == '0') ? 0 : 1;
        const int inlaunch_max = (il && il[0] == 'a') ? 16 : 4;
        bool fits = true;
        for (int i = 0; i < G; ++i) fits = fits && (size_t)grp_at(a, i).splitk_ws_bytes >= need;
        if (inlaunch && splitk <= inlaunch_max && fits) {
            const long red_off = own_tickets ? -2L : (long)((slabs + 255) / 256 * 256);
            for (int i = 0; i < G; ++i) {
                grp.a[i] = grp_at(a, i);
                // ticket words: the caller's zeroed ones, or the tail of the scratch, zeroed here (a kernel node: see op_fill_zero)
                if (!own_tickets) TRY(op_fill_zero((char*)grp.a[i].splitk_ws + red_off, twords * sizeof(int), s));
            }
            char ex[40]; snprintf(ex, sizeof(ex), " splitk%d in-launch", splitk);
            prof_igemm(a, ex);
            sym();
            LAUNCH(tag, (igemm8_kernel<NI, MODE>), dim3(per * G), dim3(512), smem, s, grp, per, ntm, ntn, splitk, order, red_off);
            return 0;
        }
        for (int i = 0; i < G; ++i) grp.a[i] = splitk_slab_desc(grp_at(a, i));
        char ex[32]; snprintf(ex, sizeof(ex), " splitk%d", splitk);
        prof_igemm(a, ex);
        sym();
        LAUNCH(tag, (igemm8_kernel<NI, MODE>), dim3(per * G), dim3(512), smem, s, grp, per, ntm, ntn, splitk, order, (long)-1);
        return launch_splitk_finish(a, splitk, s);
    }
    for (int i = 0; i < G; ++i) grp.a[i] = grp_at(a, i);
    prof_igemm(a, "");
    sym();
    LAUNCH(tag, (igemm8_kernel<NI, MODE>), dim3(per * G), dim3(512), smem, s, grp, per, ntm, ntn, 1, order, (long)-1);
    return 0;
}

}  // namespace

// number of K splits op_igemm will use for this problem (1 = none); callers size splitk_ws = factor*M*Nout*4 bytes

int igemm_set_wide(int mode) {
    if (mode < -1 || mode > 2) return 1;
    g_igemm8_mode = mode;            // -1: back to CTRL_IGEMM8 / the default
    return 0;
}

int igemm_set_order(const char* spec) {
    const OrderSpec sp = parse_order(spec);
    if (sp.kind < 0) return 1;
    g_order_spec = sp;
    return 0;
}

void igemm_tile_of(int bid, int ntm, int ntn, int mode, int group, int* tile_m, int* tile_n) {
    tileorder::tile_of(bid, ntm, ntn, tileorder::make_order(mode, group), tile_m, tile_n);
}

// bytes of split-K scratch that let op_igemm use its best form for `sk` splits: the row slabs of the finish-kernel form, or (wide
// tiles, 2..4 splits) the tile-shaped fragment slabs + ticket words of the in-launch reduction, whichever is larger
size_t igemm_splitk_ws_bytes(const IGemmArgs& a, int sk) {
    const size_t rows = (size_t)sk * a.M * a.Nout * sizeof(float);
    const int bn = (a.Nout % 320 == 0) ? 320 : 256;
    const size_t tiles = (size_t)((a.M + 255) / 256) * ((a.Nout + bn - 1) / bn);
    // (fragment slabs + ticket words for callers that do not hand in their own: one word per tile for 2..4 splits, five for the
    //  two-level reduction of 5..16 splits)
    const size_t frag = (tiles * sk * 256 * bn * sizeof(float) + 255) / 256 * 256 + tiles * (sk > 4 ? 5 : 1) * sizeof(int);
    return rows > frag ? rows : frag;
}

// ticket words (one per 256-row output tile) of the in-launch reduction for `sk` splits; 0 when that form does not apply
size_t igemm_splitk_ticket_words(const IGemmArgs& a, int sk) {
    if (sk < 2 || sk > 16) return 0;
    const int bn = (a.Nout % 320 == 0) ? 320 : 256;
    return (size_t)((a.M + 255) / 256) * ((a.Nout + bn - 1) / bn) * (sk > 4 ? 5 : 1);
}

int igemm_splitk_factor(const IGemmArgs& a) {
    if (a.geglu || a.nseg != 1 || a.seg[0].fmt != SEG_ROW || a.Nout % 64 != 0 || a.Cin % 32 != 0 || a.mode == IG_TEMPORAL) return 1;
    const int bn = (a.Nout % 320 == 0) ? 320 : 256;
    if (a.Nout % bn != 0) return 1;
    const long tiles = (long)((a.M + 255) / 256) * (a.Nout / bn);
    const int nk32 = a.Ktot / 32;
    if (tiles >= 160 || nk32 < 64) return 1;
    int sk = (int)((256 + tiles - 1) / tiles);
    if (sk > 16) sk = 16;
    while (sk > 1 && nk32 / sk < 16) --sk;
    // The kernels give split s the k-tiles [s * per, (s + 1) * per) with per = ceil(nk / sk) -- rounded up to even for the paired
    // split-operand walk, which never cuts a (hi, lo) pair -- in THEIR k-tile depth: 64 for the 8-phase kernel, 32 for the ring kernel.
    // Pick sk such that the last split still has work in the depth of the kernel that will run (640 -> 640 3x3 at 24
    // tiles: nk64 = 90, sk = 11, per = 9 left split 10 empty -- an idle workgroup whose prologue DMA started at k = Ktot)
    const int depth = can_use8(a) ? 64 : 32;
    const int nk = a.Ktot / depth;
    auto per_of = [&](int f) { int p = (nk + f - 1) / f; if (a.a_split == 2) p = (p + 1) & ~1; return p; };
    while (sk > 1 && (long)(sk - 1) * per_of(sk) >= nk) --sk;
    return sk;
}
namespace {

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int NSTAGE, int MODE>
int launch_cfg(const IGemmArgs& a, hipStream_t s) {
    if (can_swap(a)) return launch_cfg2<BM, BN, BK, WAVES_M, WAVES_N, NSTAGE, MODE, true>(a, s);
    return launch_cfg2<BM, BN, BK, WAVES_M, WAVES_N, NSTAGE, MODE, false>(a, s);
}

// first column of the transposed last segment of a MIXED segment list on the vector epilogue (0 = not such a problem): every tile must
// lie in one segment, so only tile widths that divide it may be chosen
int mixed_boundary(const IGemmArgs& a) {
    if (a.nseg < 2 || a.seg[a.nseg - 1].fmt != SEG_TRANSPOSED || !can_swap(a)) return 0;
    return a.seg[a.nseg - 1].col_begin;
}

template <int MODE>
int dispatch(const IGemmArgs& a, hipStream_t s) {
    const bool bk64 = (a.Cin % 64) == 0;
    const int mb = mixed_boundary(a);
    auto al = [&](int bn) { return mb == 0 || mb % bn == 0; };
    // Tile choice: 256x128 (8 waves) when the grid still over-fills the 256 CUs, 128x128 otherwise, 64x64 for small
    // problems.  N extents that are not multiples of 128 pay padded columns, so the waste is weighed against tile size.
    // (a grouped launch fills the chip with all its problems: the tile is sized for the group, unless the caller wants every problem
    // computed with the tile it would get alone -- ctrl_group_launches(2), bit-identical to one-by-one launches)
    const long gmul = group_tiles_as_alone() ? 1 : grp_count();
    auto tiles = [&](int bm, int bn) { return gmul * (long)((a.M + bm - 1) / bm) * ((a.Nout + bn - 1) / bn); };
    auto eff = [&](int bn) { return (double)a.Nout / (double)(((a.Nout + bn - 1) / bn) * bn); };
    // 256x256 (8 waves, 128x64 per wave): fewest LDS bytes per FLOP -- the limiter of the smaller tiles on this chip
    // split-K: small-M / long-K problems (the ControlNet's low-resolution 3x3 convs) leave most CUs idle with 256-row
    // tiles and run far below MFMA speed with small tiles; splitting the reduction fills the chip with the wide tile
    {
        const int sk = (a.splitk_ws && can_swap(a)) ? igemm_splitk_factor(a) : 1;
        if (sk > 1 && (size_t)a.splitk_ws_bytes >= (size_t)sk * a.M * a.Nout * sizeof(float) && (((uintptr_t)a.splitk_ws & 15) == 0)) {
            if (can_use8(a) && a.Nout % 320 == 0) return launch8<5, MODE>(a, s, sk);
            if (can_use8(a) && a.Nout % 256 == 0) return launch8<4, MODE>(a, s, sk);
            if (a.Nout % 320 == 0) return launch_cfg2<256, 320, 32, 2, 4, 4, MODE, true>(a, s, sk);
            return launch_cfg2<256, 256, 32, 2, 4, 4, MODE, true>(a, s, sk);
        }
    }
    const char* const forced = policy_raw(P_IGEMM_FORCE);
    if (const char* f = forced) {      // tile experiments (tools/tile_experiment.py), row outputs only
        if (can_swap(a) && MODE == IG_ROWS && a.nseg == 1 && !a.geglu) {
            // round 5, short launches of the small-M chain (tools/gemm_order_bench small): 4-wave tiles at three workgroups per CU
            if (!strcmp(f, "128x128x32")) return launch_cfg2<128, 128, 32, 2, 2, 3, MODE, true>(a, s);
            if (!strcmp(f, "128x64x64") && bk64) return launch_cfg2<128, 64, 64, 2, 2, 3, MODE, true>(a, s);
            if (!strcmp(f, "64x64x64") && bk64) return launch_cfg2<64, 64, 64, 2, 2, 3, MODE, true>(a, s);
            if (!strcmp(f, "128x128x64") && bk64) return launch_cfg2<128, 128, 64, 2, 4, 4, MODE, true>(a, s);
            if (!strcmp(f, "wide") && can_use8(a)) {
                if (a.Nout % 320 == 0) return launch8<5, MODE>(a, s);
                if (a.Nout % 256 == 0) return launch8<4, MODE>(a, s);
            }
        }
        // (a mixed-layout problem -- Q | K row-major + V^T transposed -- only takes a forced tile whose width divides the transposed
        // segment's first column: a tile must lie in one segment, the epilogue's choice is workgroup-uniform)
        if (can_swap(a) && MODE == IG_ROWS) {
            if (!strcmp(f, "128x256") && al(256)) return launch_cfg2<128, 256, 32, 2, 4, 3, MODE, true>(a, s);
            if (!strcmp(f, "256x128") && al(128)) return launch_cfg2<256, 128, 32, 4, 2, 3, MODE, true>(a, s);
            if (!strcmp(f, "4w256x128") && al(128)) return launch_cfg2<256, 128, 32, 2, 2, 3, MODE, true>(a, s);      // 4 waves x 256 registers, two workgroups per CU
            if (!strcmp(f, "4w128x256") && al(256)) return launch_cfg2<128, 256, 32, 1, 4, 3, MODE, true>(a, s);
        }
    }
    // Epilogue-heavy token GEMMs at large M with a SHORT k-loop (measured, tools/tile_experiment.py -> profiles/r02_tile_experiment.log,
    // re-measured against the 8-phase kernel in round 4, profiles/r04_step_old_vs_8phase.txt): two resident workgroups per CU let
    // one's epilogue -- the GEGLU math, or the HBM-bound fp32 residual read + fp32 master + fp16 mirror write of a stream update --
    // run under the other's k-loop, which the one-workgroup-per-CU wide tile cannot.  GEGLU 512 -> 4096 at M = 131072: 653 (256x128
    // pair) vs 635 TFLOP/s (8-phase); stream update K = 320: 273 vs 221.  From K = 2048 on the 8-phase loop wins (815 vs 740).
    const bool force8 = igemm8_mode() == 2;
    if (!force8 && MODE == IG_ROWS && a.M >= 65536 && can_swap(a) && a.nseg == 1 && a.seg[0].fmt == SEG_ROW && a.Nout % 256 == 0 && al(256)) {
        const bool f32_stream = a.seg[0].dtype == DT_F32 || (a.res && a.res_f32);      // fp32 rows written and / or read per element
        if (a.geglu) return launch_cfg2<256, 128, 32, 4, 2, 3, MODE, true>(a, s);
        if (f32_stream && a.Ktot <= 1024) return launch_cfg2<128, 256, 32, 2, 4, 3, MODE, true>(a, s);
        // fp16 stream updates (the adapter's token stream since round 4) with a k-loop of <= 8 k-tiles: the residual read + store of
        // the epilogue still outweighs the loop -- M131072 N512 K320 + residual: 0.164 ms (8-phase) vs 0.128 ms (this pair tile),
        // -0.12 ms per SDXL step, -0.27 ms fused (one call, gpurun_out/r4p); CTRL_SHORTK_PAIR=0 switches it off
        const bool pair16 = !policy_is0(P_SHORTK_PAIR);
        if (pair16 && a.res && a.Ktot <= 512) return launch_cfg2<128, 256, 32, 2, 4, 3, MODE, true>(a, s);
    }
    // the 8-phase wide tiles wherever the grid fills the chip with them (1.3-1.5x the BK = 32 ring kernel on plain epilogues and
    // long k-loops: convolutions 543 -> 671 TFLOP/s as a class)
    if (can_use8(a) && forced == nullptr) {
        if (a.Nout % 320 == 0 && (tiles(256, 320) >= 160 || force8) && !a.geglu && al(320)) return launch8<5, MODE>(a, s);
        if (a.Nout % 256 == 0 && (tiles(256, 256) >= 200 || force8) && al(256)) return launch8<4, MODE>(a, s);
    }
    // (vector-epilogue variant only: the scalar-epilogue one does not fit the register file at this tile size)
    if (tiles(256, 256) >= 200 && eff(256) > 0.9 && can_swap(a) && al(256)) return launch_cfg2<256, 256, 32, 2, 4, 4, MODE, true>(a, s);
    // N = 320 / 640 / 960 / 1280 / 1920 / 3840 (every conv and QKV width of the path): 256x320 tile, 128x80 per wave
    if (a.Nout % 320 == 0 && tiles(256, 320) >= 160 && can_swap(a) && !a.geglu && al(320)) return launch_cfg2<256, 320, 32, 2, 4, 4, MODE, true>(a, s);
    if (!bk64) {
        if (tiles(128, 128) >= 192 && al(128)) return launch_cfg<128, 128, 32, 2, 2, 3, MODE>(a, s);
        return launch_cfg<64, 64, 32, 2, 2, 3, MODE>(a, s);
    }
    // (a 256x128x64 tile at two workgroups per CU stood here: every instantiation spilled inside its MFMA loop -- 128 registers do
    // not hold a 64-deep fragment set -- and the shapes it served now run on the 8-phase tiles; removed in round 4)
    // Short launches of the small-M chains (the ControlNet at b = 8, the 64^2 .. 8^2 adapter levels: a few waves of tiles, k-loops of
    // 5 .. 40 k-tiles): what counts is how many workgroups a CU can interleave, not the tile's LDS traffic per FLOP.  Measured on the
    // path's shapes (tools/gemm_order_bench small -> profiles/r05_gemm_small_tiles.txt): the 4-wave 128 x 128 x 32 tile at three
    // workgroups per CU beats the 8-wave 128 x 128 x 64 tile (one per CU) by 10-18 % from ~500 tiles on, the 64 x 64 x 64 tile
    // (three per CU) by 10-20 % below that for N <= 1280.  CTRL_SMALL_TILES=0: the round-4 choice.
    const bool small_tiles = !policy_is0(P_SMALL_TILES);
    if (small_tiles && MODE == IG_ROWS && can_swap(a) && !a.geglu) {
        if (tiles(128, 128) >= 512 && eff(128) > 0.8 && al(128)) return launch_cfg2<128, 128, 32, 2, 2, 3, MODE, true>(a, s);
        if (tiles(128, 128) < 512 && a.Nout <= 1280 && al(64)) return launch_cfg2<64, 64, 64, 2, 2, 3, MODE, true>(a, s);
    }
    if (tiles(128, 128) >= 192 && eff(128) > 0.8 && al(128)) return launch_cfg<128, 128, 64, 2, 4, 4, MODE>(a, s);
    if (tiles(128, 64) >= 192) return launch_cfg<128, 64, 64, 2, 2, 3, MODE>(a, s);
    return launch_cfg<64, 64, 64, 2, 2, 3, MODE>(a, s);
}

}  // namespace

static int igemm_validate(const IGemmArgs& a);
static int igemm_dispatch(const IGemmArgs& a, hipStream_t s) {
    if (a.mode == IG_CONV2D) return dispatch<IG_CONV2D>(a, s);
    if (a.mode == IG_TEMPORAL) return dispatch<IG_TEMPORAL>(a, s);
    return dispatch<IG_ROWS>(a, s);
}
int op_igemm(const IGemmArgs& a, hipStream_t s) {
    if (t_collect) {                                 // lock-step replay of sibling blocks: deposited, launched by the collector's flush()
        int rc = 0;
        const int i = t_collect->slot(OpCollector::IGEMM, s, &rc);
        if (i < 0) return rc;
        t_collect->ig[i] = a;
        return 0;
    }
    IGemmArgs b = a;
    if (!b.nonfinite && range_check_on()) {          // debug aid: every fp16 value the epilogue writes is tested for inf / nan
        b.nonfinite = range_flag();
        CTRL_CHECK(b.nonfinite != nullptr, "igemm: the range check is on but its flag word could not be allocated");
    }
    TRY(igemm_validate(b));
    return igemm_dispatch(b, s);
}

// what has to agree for two problems to share a launch: every field the dispatcher, the launch geometry and the kernels' uniform
// branches depend on -- i.e. everything but the addresses, of which only presence and 16-byte alignment matter
static bool igemm_same_form(const IGemmArgs& x, const IGemmArgs& y) {
    auto pk = [](const void* p) { return p ? 1 + (int)((uintptr_t)p & 15) : 0; };
#define SAME(f) (x.f == y.f)
#define SAMEP(f) (pk((const void*)x.f) == pk((const void*)y.f))
    bool ok = SAME(lda) && SAME(mode) && SAME(Cin) && SAME(taps) && SAME(Hin) && SAME(Win) && SAME(Hout) && SAME(Wout) && SAME(stride) &&
              SAME(up) && SAME(F) && SAME(HW) && SAME(t_pad) && SAME(M) && SAME(Nout) && SAME(Ktot) && SAME(rowvec_ld) && SAME(rows_per_img) &&
              SAME(ldres) && SAME(scale) && SAME(geglu) && SAME(nseg) && SAME(act) && SAME(res_f32) && SAME(a_split) &&
              SAME(splitk_ws_bytes) && SAME(ld16) && SAME(ld_blend) && SAME(blend_f32) && SAME(out16_lo_off) && SAME(scale2) &&
              SAME(scale2_from) && SAME(scale2_to) && SAME(res_up) &&
              SAMEP(A) && SAMEP(W) && SAMEP(bias) && SAMEP(rowvec) && SAMEP(res) && SAMEP(splitk_ws) && ((x.splitk_tickets == nullptr) == (y.splitk_tickets == nullptr)) && SAMEP(out16) && SAMEP(blend_mix) &&
              SAMEP(blend_x) && SAMEP(nonfinite);
    for (int i = 0; ok && i < x.nseg; ++i)
        ok = SAME(seg[i].ld) && SAME(seg[i].col_begin) && SAME(seg[i].ncols) && SAME(seg[i].fmt) && SAME(seg[i].dtype) && SAME(seg[i].L) &&
             SAMEP(seg[i].out) && SAMEP(seg[i].img_map);
#undef SAME
#undef SAMEP
    return ok;
}

int op_igemm_group(const IGemmArgs* a, int n, hipStream_t s) {
    CTRL_CHECK(a && n >= 1 && n <= kMaxIGemmGroup, "igemm_group: 1..4 problems");
    bool same = n > 1 && group_launches_enabled();
    for (int i = 1; same && i < n; ++i) same = igemm_same_form(a[0], a[i]);
    if (!same) {
        for (int i = 0; i < n; ++i) TRY(op_igemm(a[i], s));
        return 0;
    }
    IGemmArgs b[kMaxIGemmGroup];
    for (int i = 0; i < n; ++i) {
        b[i] = a[i];
        if (!b[i].nonfinite && range_check_on()) {
            b[i].nonfinite = range_flag();
            CTRL_CHECK(b[i].nonfinite != nullptr, "igemm: the range check is on but its flag word could not be allocated");
        }
        TRY(igemm_validate(b[i]));
    }
    t_grp = b; t_grp_n = n;            // the launch functions pick the siblings up from here
    const int rc = igemm_dispatch(b[0], s);
    t_grp = nullptr; t_grp_n = 1;
    return rc;
}

static int igemm_validate(const IGemmArgs& a) {
    CTRL_CHECK(a.M > 0 && a.Nout > 0 && a.Ktot > 0, "igemm: empty problem");
    CTRL_CHECK(a.Cin % 32 == 0, "igemm: Cin must be a multiple of 32 (got " + std::to_string(a.Cin) + ")");
    CTRL_CHECK(a.Ktot == a.taps * a.Cin, "igemm: Ktot != taps*Cin");
    CTRL_CHECK(a.a_split != 2 || ((a.Cin / 2) % 64 == 0 && a.mode != IG_TEMPORAL), "igemm: the paired split-operand walk needs Cin/2 % 64 == 0");
    CTRL_CHECK(a.lda % 8 == 0, "igemm: lda must be a multiple of 8 (16-byte vector loads)");
    CTRL_CHECK(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "igemm: A/W must be 16-byte aligned");
    CTRL_CHECK(a.nseg >= 1 && a.nseg <= 3, "igemm: nseg must be 1..3");
    CTRL_CHECK(!a.geglu || (a.Nout % 32) == 0, "igemm: GEGLU needs Nout % 32 == 0");
    CTRL_CHECK(a.scale2_from == 0 || (a.scale2_from % 16 == 0 && a.scale2_to % 16 == 0 && (a.scale2_to == 0 || a.scale2_to > a.scale2_from) &&
                                      !a.geglu && !a.splitk_ws),
               "igemm: the scale2 column range must be bounded by multiples of 16 (no GEGLU, no split-K)");
    for (int i = 0; i + 1 < a.nseg; ++i)
        CTRL_CHECK(a.seg[i].col_begin < a.seg[i + 1].col_begin, "igemm: segments must be listed in ascending column order");
    CTRL_CHECK(!a.out16 || (a.nseg == 1 && a.seg[0].fmt == SEG_ROW && can_swap(a)), "igemm: the fp16 mirror needs a single aligned row-major output");
    CTRL_CHECK(!a.blend_mix || (a.blend_x && a.nseg == 1 && a.seg[0].fmt == SEG_ROW && !a.geglu && can_swap(a)),
               "igemm: the blend fold needs a single aligned row-major output and an aligned blend operand");
    CTRL_CHECK(a.res_up == 0 || a.res_up == 1 || (a.res_up == 2 && a.res && a.mode == IG_CONV2D && a.Hout % 2 == 0 && a.Wout % 2 == 0 &&
                                                   a.nseg == 1 && a.seg[0].fmt == SEG_ROW),
               "igemm: an up-sampled residual (res_up = 2) needs a conv2d problem with even output size and one row-major output");
    for (int i = 0; i < a.nseg; ++i) {
        CTRL_CHECK(a.seg[i].col_begin % 16 == 0, "igemm: segment boundary must be a multiple of 16");
        CTRL_CHECK(a.seg[i].out != nullptr, "igemm: null segment output");
    }
    if (can_swap(a)) {
        // the vector epilogue addresses every row-major operand with 32-bit unsigned ELEMENT offsets off a scalar base (igemm_epilogue:
        // res_fetch8, the piece stores, the blend operand, the fp16 mirror): each of them must span < 2^32 elements
        const double lim = 4294967296.0, M = (double)a.M;
        for (int i = 0; i < a.nseg; ++i)
            CTRL_CHECK(a.seg[i].fmt != SEG_ROW || M * (double)a.seg[i].ld < lim, "igemm: a row-major output spans 2^32 elements or more");
        CTRL_CHECK(!a.res || M * (double)a.ldres < lim, "igemm: the residual operand spans 2^32 elements or more");
        CTRL_CHECK(!a.out16 || M * (double)a.ld16 + (double)a.out16_lo_off < lim, "igemm: the fp16 mirror spans 2^32 elements or more");
        CTRL_CHECK(!a.blend_mix || M * (double)a.ld_blend < lim, "igemm: the blend operand spans 2^32 elements or more");
    }
    if (a.mode == IG_CONV2D) {
        CTRL_CHECK(a.taps == 9 || a.taps == 1, "igemm conv2d: taps must be 1 or 9");
        CTRL_CHECK((a.up == 1 || a.up == 2) && (a.stride == 1 || a.stride == 2), "igemm conv2d: up/stride must be 1|2");
    } else if (a.mode == IG_TEMPORAL) {
        CTRL_CHECK(a.taps == 3 && a.F > 0 && a.HW > 0, "igemm temporal: taps must be 3");
    } else {
        CTRL_CHECK(a.taps == 1, "igemm rows: taps must be 1");
    }
    return 0;
}

int op_linear(const half_t* A, long lda, const half_t* W, const float* bias, half_t* out, long ldo,
              int M, int N, int K, const half_t* res, long ldres, hipStream_t s) {
    IGemmArgs g = {};
    g.A = A; g.lda = lda; g.mode = IG_ROWS; g.Cin = K; g.taps = 1;
    g.W = W; g.M = M; g.Nout = N; g.Ktot = K;
    g.bias = bias; g.res = res; g.ldres = ldres; g.scale = 1.f;
    g.nseg = 1;
    g.seg[0] = IGemmSeg{out, ldo, 0, N, SEG_ROW, DT_F16, 1, 0};
    return op_igemm(g, s);
}
