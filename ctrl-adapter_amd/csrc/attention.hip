// Flash attention for gfx950 (v_mfma_f32_32x32x16_f16), fp32 online softmax.
//
// Replaces F.scaled_dot_product_attention inside diffusers' Attention/AttnProcessor2_0 as reached from
//   * the adapter's spatial BasicTransformerBlock (model/adapter_spatial_temporal.py:108-116, called :271):
//     heads = C/64, head_dim 64, L = H*W up to 16384 (SDXL 128x128), cross-attention Lk = 77
//   * the ControlNet's Transformer2DModel blocks (controlnet/controlnet.py:371-391,410-424):
//     8 heads of 40/80/160, L = 4096/1024/256/64, cross-attention Lk = 77.
//
// Structure (one workgroup = 128 queries of one (batch, head); 4 wavefronts x 32 queries):
//   S^T = K.Q^T   -- "swapped" product: the MFMA result column is the query, so every lane owns ONE
//                    query and 32 of the 64 keys of the tile; row max / row sum need a single exchange
//                    with lane^32 and the rescale factor is a per-lane scalar.
//   O^T += V^T.P^T -- the P^T B-operand is exactly the lane's own exponentiated scores (no cross-lane
//                    movement) because K rows are fetched from LDS through a bit-2<->bit-3 swapped row
//                    index; V is supplied transposed ([C][tokens], written that way by the QKV GEMM
//                    epilogue) so its A-operand is one ds_read_b128 of 8 consecutive keys.
//   K / V^T tiles (64 keys) stream global -> LDS through the asynchronous LDS-DMA (global_load_lds_dwordx4) into a
//   3-deep ring: two tiles are always in flight, a counted s_waitcnt vmcnt + one raw s_barrier per tile; linear LDS
//   tiles with an XOR chunk swizzle on the source address and on the fragment read (bank-conflict-free b128 reads).
//   Contract: the V^T buffer's pad columns [Lk, roundup8(Lk)) must hold finite values (the producer zero-fills).
#include "ops.h"
#include <cstdlib>

namespace {

__device__ __forceinline__ int krow_perm(int i) {   // swap bits 2 and 3 (identity on bits 0,1,4)
    return (i & 0x13) | ((i & 4) << 1) | ((i & 8) >> 1);
}

// XOR chunk swizzles of the linear LDS tiles (applied to the staging SOURCE address and to the fragment read; the
// LDS-DMA destination itself is wave-uniform base + lane*16).  128-byte rows: 2 rows per 256-B bank line -> 3 bits of
// (row>>1); 192/320-byte rows (12/20 chunks): the row start advances 4 slots mod 16 -> 2 bits of (row>>2).
template <int ROW_CHUNKS>
__device__ __forceinline__ int tile_swz(int row) {
    return ROW_CHUNKS == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3);
}

template <int D, int NW, bool BATCH, bool FOLD>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 4 : 1) void flash_attn_kernel(AttnGroup kargs, int per, const half_t* zeros) {
    const int gp = __builtin_amdgcn_readfirstlane(blockIdx.x / per);
    const int gbid = blockIdx.x - gp * per;
    const AttnArgs a = ATTN_GROUP_ARGS(gp);
    constexpr int DP = (D + 31) / 32 * 32;      // padded head dim (zero filled): 64, 64, 96, 160
    constexpr int KS = (D + 15) / 16;           // k-steps of the 32x32x16 MFMA for QK^T (40 -> 3, 80 -> 5: no all-zero steps)
    constexpr int DB = DP / 32;                 // 32-row output blocks of O^T
    constexpr int KCPR = DP / 8;                // 16-B chunks per K-tile row
    constexpr int PASSES = DP / (8 * NW);       // staging passes per tile: NW KiB each (K tile = V^T tile = 64*DP halfs)
    static_assert(DP % (8 * NW) == 0, "tile bytes must be a multiple of the workgroup's staging pass");
    constexpr int NSTAGE = 3;                   // LDS ring depth: 2 tiles in flight
    constexpr int LPT = 2 * PASSES;             // global_load_lds per lane per tile
    constexpr int TILE = 64 * DP;               // halfs per K (or V^T) tile

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    half_t* Ks = (half_t*)smem_raw;             // [NSTAGE][64][DP]
    half_t* Vs = Ks + NSTAGE * TILE;            // [NSTAGE][DP][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform: LDS-DMA destinations and per-wave offsets stay in SGPRs)
    const int hi = lane >> 5, lq = lane & 31;
    // XCD-aware work map (1-D grid): workgroup id -> XCD id&7 (observed dispatch order); every XCD owns whole (batch, head)
    // pairs p = x, x+8, x+16, ... and walks their query tiles back to back, so the workgroups resident on one XCD at any
    // time stream the SAME K/V tiles through that XCD's private L2 (K+V of one pair at L = 16384 is 4 MiB = one L2).
    const int qtiles = (a.Lq + NW * 32 - 1) / (NW * 32);
    int pair, qt;
    if (!attn_work_map(gbid, qtiles, a.B * a.heads, &pair, &qt)) return;      // (ops.h: whole pairs per XCD, the last partial round dealt over all XCDs)
    const int b = pair / a.heads, h = pair - b * a.heads;
    const int q0 = qt * (NW * 32) + wave * 32;
    const int Lq = a.Lq, Lk = a.Lk;
    // softmax in the exp2 domain.  FOLD: K rows arrive pre-multiplied by scale*log2(e) (ctrl_attn_desc::k_prescaled), so the
    // MFMA output needs no scaling, and the running maximum is subtracted by INITIALISING the QK^T accumulator with -m
    // (a lane owns one query = one column of S^T): the MFMA pipe does the subtraction and the loop loses its 32 v_fma per
    // tile -- it is VALU-bound at head_dim 64 (~140 VALU against 16 MFMAs per 64-key tile)
    const float c = FOLD ? 1.0f : a.scale * 1.4426950408889634f;

    const h8 hzero = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- Q fragments (B operand: col = query, k = hi*8+j) kept in registers ----
    h8 qf[KS];
    {
        int q = q0 + lq;
        if (q > Lq - 1) q = Lq - 1;               // clamp (rows beyond Lq are computed but never stored)
        const half_t* qp = (const half_t*)a.Q + ((size_t)b * Lq + q) * a.ldq + (size_t)h * D;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = ks * 16 + hi * 8;
            qf[ks] = (d0 < D) ? *(const h8*)(qp + d0) : hzero;
        }
        // Retire the Q loads HERE: a pending ordinary load at loop entry makes hipcc put s_waitcnt vmcnt(0) in front of
        // the first MFMA of every iteration, which would drain the LDS-DMA ring each tile.
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
    }

    const int kb = (a.kvB == 1) ? 0 : b;     // K/V shared by every batch (broadcast encoder states)
    const half_t* Kbase = (const half_t*)a.K + (size_t)kb * Lk * a.ldk + (size_t)h * D;
    const half_t* Vbase = (const half_t*)a.Vt + ((size_t)kb * a.heads + h) * D * (size_t)a.Lkpad;

    // ---- per-lane staging coordinates (constant over tiles) ----
    int k_row[PASSES], v_key[PASSES];
    size_t k_off[PASSES], v_off[PASSES];
    bool k_dok[PASSES], v_dok[PASSES];
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        const int ci = (i * NW + wave) * 64 + lane;           // chunk index inside the tile
        const int kr = ci / KCPR, kp = ci - kr * KCPR;
        const int ksrc = kp ^ tile_swz<KCPR>(kr);
        k_row[i] = kr;
        k_dok[i] = ksrc * 8 < D;
        k_off[i] = (size_t)kr * a.ldk + ksrc * 8;
        const int vr = ci >> 3, vp = ci & 7;
        const int vsrc = vp ^ tile_swz<8>(vr);
        v_key[i] = vsrc * 8;
        v_dok[i] = vr < D;
        v_off[i] = (size_t)vr * a.Lkpad + vsrc * 8;
    }
    typedef const void __attribute__((address_space(1)))* gptr_t;
    typedef void __attribute__((address_space(3)))* lptr_t;
    auto stage = [&](int t, int slot) {
        const int kt0 = t * 64;
        half_t* Kb = Ks + slot * TILE;
        half_t* Vb = Vs + slot * TILE;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const bool ok = k_dok[i] && (kt0 + k_row[i] < Lk);
            const half_t* src = ok ? (Kbase + (size_t)kt0 * a.ldk + k_off[i]) : zeros;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Kb + (i * NW + wave) * 512), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            // whole 8-key chunks at or beyond Lk come from the zero page; the chunk straddling Lk relies on the
            // V^T pad columns being finite (the producer zero-fills them) -- their probabilities are exactly 0
            const bool ok = v_dok[i] && (kt0 + v_key[i] < Lk);
            const half_t* src = ok ? (Vbase + kt0 + v_off[i]) : zeros;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Vb + (i * NW + wave) * 512), 16, 0, 0);
        }
    };

    f16v o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = FOLD ? 0.f : -1e30f, l_run = 0.f;

    const int ntiles = (Lk + 63) / 64;
#pragma unroll
    for (int t = 0; t < NSTAGE - 1; ++t)
        if (t < ntiles) stage(t, t);

    // fragment read offsets (halfs) inside a tile
    const int krow = krow_perm(lq);
    int k_rd[2], k_sw[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) { const int r = mb * 32 + krow; k_rd[mb] = r * DP; k_sw[mb] = tile_swz<KCPR>(r); }
    int v_rd[DB], v_sw[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) { const int r = db * 32 + lq; v_rd[db] = r * 64; v_sw[db] = tile_swz<8>(r); }

    int cur = 0;
    for (int t = 0; t < ntiles; ++t) {
        if (t + NSTAGE - 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // Every fragment read of tile t-1 is RETIRED before this barrier (round 6): the workgroup re-stages that tile's ring slot right
        // behind it, and hipcc, left alone, schedules the barrier ABOVE the tile's last MFMA and the lgkmcnt wait of its operand -- the
        // read is then still in flight when another wave's LDS-DMA overwrites the slot.  Found as a 3-8 % per-replay flake of the captured,
        // multi-lane adapter forward (one wave's 32 queries of one head slightly off; tools/diag/graph_replay_stress.py); never seen with
        // one kernel at a time on the chip.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // tile t visible to every wave; the slot of tile t-1 is free
        if (t + NSTAGE - 1 < ntiles) {
            int ns = cur + NSTAGE - 1;
            if (ns >= NSTAGE) ns -= NSTAGE;
            stage(t + NSTAGE - 1, ns);
        }
        const half_t* Kb = Ks + cur * TILE;
        const half_t* Vb = Vs + cur * TILE;

        // ---- S^T = K . Q^T  (two 32-key blocks); all K and V^T fragment reads of the tile are issued up front so the
        //      LDS latency is paid once (K) or hidden under the softmax (V^T) ----
        f16v sacc[2];
        const float acc0 = FOLD ? -m_run : 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[mb][r] = acc0;
        h8 vfr[4][DB];
        if constexpr (BATCH) {
            h8 kfr[KS][2];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) kfr[ks][mb] = *(const h8*)(Kb + k_rd[mb] + (((ks * 2 + hi) ^ k_sw[mb]) << 3));
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                for (int db = 0; db < DB; ++db) vfr[c4][db] = *(const h8*)(Vb + v_rd[db] + (((c4 * 2 + hi) ^ v_sw[db]) << 3));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
                    sacc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfr[ks][mb], qf[ks], sacc[mb], 0, 0, 0);
        } else {
            // low-register form (2 workgroups of 8 waves per CU): fragments are fetched right before their MFMAs;
            // four resident waves per SIMD hide the LDS latency instead of a register-resident batch
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const h8 kf = *(const h8*)(Kb + k_rd[mb] + (((ks * 2 + hi) ^ k_sw[mb]) << 3));
                    sacc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], sacc[mb], 0, 0, 0);
                }
        }
        // register r of block mb holds key  kt0 + 32*mb + 16*(r>>3) + 8*hi + (r&7)  (see krow_perm)
        const int kt0 = t * 64;
        if (kt0 + 64 > Lk) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt0 + mb * 32 + ((r >> 3) << 4) + hi * 8 + (r & 7);
                    if (key >= Lk) sacc[mb][r] = -1e30f;
                }
        }
        // ---- online softmax (exp2 domain) ----
        float mx = sacc[0][0];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = vmaxf(mx, sacc[mb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float psum = 0.f;
        h8 pf[2][2];
        if constexpr (FOLD) {
            // sacc = s - m_run already; mx > 0 <=> the row maximum moved (first tile: m_run = 0 stands for "none yet" and
            // the update is forced so that an all-negative first tile is referenced to its own maximum)
            // The reference point is only moved when a score exceeds it by more than 2^4: p <= 16 is as exact in fp16 / fp32 as
            // p <= 1, and with 64 queries per wave SOME lane sees a new maximum on nearly every tile of a long sequence -- the
            // exact test ran the 64-instruction rescale almost every tile (round 3: +9..17 % on the long-sequence kernel)
            if (t == 0 || __any(mx > 4.f)) {        // wave-uniform
                const float d = (t == 0) ? mx : fmaxf(mx, 0.f);
                const float alpha = (t == 0) ? 1.f : __builtin_amdgcn_exp2f(-d);   // t == 0: l_run = o = 0 (and 2^-d may overflow)
                l_run *= alpha;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
                m_run += d;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[mb][r] -= d;
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(sacc[mb][r]);
                    psum += p;
                    pf[mb][r >> 3][r & 7] = (half_t)p;
                }
        } else {
            const float m_new = fmaxf(m_run, mx * c);
            // rescale only when some query's running maximum moved (exact: alpha == 1 otherwise); wave-uniform branch
            if (__any(m_new != m_run)) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
                m_run = m_new;
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(sacc[mb][r] * c - m_run);
                    psum += p;
                    pf[mb][r >> 3][r & 7] = (half_t)p;
                }
        }
        l_run += psum;

        // ---- O^T += V^T . P^T ----
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const h8 vf = BATCH ? vfr[mb * 2 + s2][db]
                                        : *(const h8*)(Vb + v_rd[db] + ((((mb * 2 + s2) * 2 + hi) ^ v_sw[db]) << 3));
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[mb][s2], o[db], 0, 0, 0);
                }
            }
        cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
    }

    // ---- finalize: O[q][d] = O^T[d][q] / l ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0 + lq;
    if (q < Lq) {
        half_t* op = (half_t*)a.O + ((size_t)b * Lq + q) * a.ldo + (size_t)h * D;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // rows (r&3) + 8*(r>>2) + 4*hi of the 32x32 C/D fragment: r = 4g..4g+3 -> 4 consecutive d
                const int d0 = db * 32 + 8 * g + 4 * hi;
                if (d0 < D) {
                    h4 v = {(half_t)(o[db][4 * g] * inv), (half_t)(o[db][4 * g + 1] * inv),
                            (half_t)(o[db][4 * g + 2] * inv), (half_t)(o[db][4 * g + 3] * inv)};
                    *(h4*)(op + d0) = v;
                }
            }
    }
}

const half_t* attn_zero_page() { return (const half_t*)device_zero_page(); }
}  // namespace
// descriptors of the grouped launch being dispatched (op_flash_attn_group); [0] is the problem the dispatcher sees
thread_local const AttnArgs* t_attn_grp = nullptr;
thread_local int t_attn_grp_n = 1;
namespace {

template <int D, int NW, bool BATCH, bool FOLD>
int launch_attn2(const AttnArgs& a, hipStream_t s) {
    constexpr int DP = (D + 31) / 32 * 32;
    constexpr size_t smem = (size_t)3 * 2 * 64 * DP * sizeof(half_t);
    static bool attr_done[kMaxDevices] = {};
    const int dev = cur_device();
    if (!attr_done[dev]) {
        HIP_TRY(hipFuncSetAttribute((const void*)flash_attn_kernel<D, NW, BATCH, FOLD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_done[dev] = true;
    }
    const half_t* zeros = attn_zero_page();
    CTRL_CHECK(zeros != nullptr, "flash_attn: could not allocate the zero page");
    const int qtiles = (a.Lq + NW * 32 - 1) / (NW * 32), pairs = a.B * a.heads;
    const int per = 8 * ((pairs + 7) / 8) * qtiles, G = attn_grp_count();
    dim3 grid((unsigned)(per * G));
    PROF_WORK(G * 4.0 * a.B * a.heads * (double)a.Lq * a.Lk * a.D, G * 2.0 * a.heads * a.D * (2.0 * a.B * a.Lq + 2.0 * a.kvB * a.Lk));
    if (G > 1) prof_detail("B%d h%d D%d Lq%d Lk%d x%d", a.B, a.heads, a.D, a.Lq, a.Lk, G);
    else prof_detail("B%d h%d D%d Lq%d Lk%d", a.B, a.heads, a.D, a.Lq, a.Lk);
    prof_symbol("flash_attn_kernel<%d, %d, %s, %s>", D, NW, BATCH ? "true" : "false", FOLD ? "true" : "false");
    LAUNCH("flash_attn", (flash_attn_kernel<D, NW, BATCH, FOLD>), grid, dim3(NW * 64), smem, s, attn_grp_make(a), per, zeros);
    return 0;
}

// k_prescaled selects the folded form (see the kernel); the plain form serves callers that hand over unscaled K.
// (A "lazy running maximum" form was measured and dropped in round 2: profiles/r02_attention_lazy_max.md.)
template <int D, int NW, bool BATCH>
int launch_attn(const AttnArgs& a, hipStream_t s) {
    return a.k_prescaled ? launch_attn2<D, NW, BATCH, true>(a, s) : launch_attn2<D, NW, BATCH, false>(a, s);
}

// ---------------------------------------------------------------------------------------------
// Temporal attention: sequence = frames (F <= 32), batch = clips x pixels, head_dim 64.
// Replaces the Attention inside diffusers' TemporalBasicTransformerBlock (constructed
// model/adapter_spatial_temporal.py:120-130, called :280) after its [bF,L,C] -> [b*L,F,C] reshape.
// The F x F score matrix is tiny, so this kernel is HBM-bound: one wavefront per (clip, pixel, head) stages the K and V
// head slices of all F key frames ONCE into a wave-private LDS area (16-byte coalesced loads, each byte fetched once),
// then every lane = (query frame, 16-dim slice) reads the key / value slices from LDS (same-address broadcast) and
// keeps the F scores in registers; the result goes back in the original frame-major layout (no materialised permute).
// Q and K|V are addressed separately so that the same kernel serves a clip whose frames are sharded over GPUs
// (SURVEY.md 8e): queries = the local frames, keys / values = the all-gathered K|V rows of every rank:
//   Q  row (b*Fq + fq)*HW + p                       in Q  (ld),   head h at column h*64
//   KV row ((r*Bc + b)*Fl + fl)*HW + p, r = kf / Fl in KV (ldkv), K at column h*64, V at column C + h*64
// (unsharded: KV = Q + C, ldkv = ld, Fl = Fq = F).
// ---------------------------------------------------------------------------------------------
template <int FP>
__global__ __launch_bounds__(256) void temporal_attn_kernel(TAttnArgs a) {
    __shared__ __attribute__((aligned(16))) half_t kv_s[4][2][FP][64];      // [wave][K|V][key frame][dim]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long item = (long)blockIdx.x * 4 + wave;
    const long nitems = (long)a.Bc * a.HW * a.heads;
    const bool live = item < nitems;
    const long it = live ? item : 0;
    const int h = (int)(it % a.heads);
    const long bp = it / a.heads;
    const int p = (int)(bp % a.HW);
    const int b = (int)(bp / a.HW);
    const int C = a.heads * 64;
    const int F = a.F, Fq = a.Fq, Fl = a.Fl;
    const float c = a.scale * 1.4426950408889634f;
    const h8 hzero = {0, 0, 0, 0, 0, 0, 0, 0};
    // ---- stage K and V of every key frame: FP rows x 8 chunks of 16 B each, one chunk per lane per pass ----
#pragma unroll
    for (int i = 0; i < (FP * 8 + 63) / 64; ++i) {
        const int idx = i * 64 + lane;
        const int kf = idx >> 3, c8 = idx & 7;
        h8 kk = hzero, vv = hzero;
        if (live && kf < F) {
            const int r = kf / Fl, fl = kf - r * Fl;
            const half_t* kp = (const half_t*)a.KV + (((size_t)(r * a.Bc + b) * Fl + fl) * a.HW + p) * a.ldkv + (size_t)h * 64 + c8 * 8;
            kk = *(const h8*)kp;
            vv = *(const h8*)(kp + C);
        }
        if (kf < FP) {
            *(h8*)&kv_s[wave][0][kf][c8 * 8] = kk;
            *(h8*)&kv_s[wave][1][kf][c8 * 8] = vv;
        }
    }
    __syncthreads();
    if (!live) return;
    const int sl = lane >> 4;          // 16-dim slice of the head
    const int fl_ = lane & 15;         // query frame handled by this lane (within a pass)
    const size_t col = (size_t)h * 64 + sl * 16;
#pragma unroll
    for (int f0 = 0; f0 < FP; f0 += 16) {
        if (f0 >= Fq) break;
        const int fq = f0 + fl_;
        const bool qok = fq < Fq;
        h8 q0 = hzero, q1 = hzero;
        if (qok) {
            const half_t* qp = (const half_t*)a.Q + ((size_t)(b * Fq + fq) * a.HW + p) * a.ld + col;
            q0 = *(const h8*)qp;
            q1 = *(const h8*)(qp + 8);
        }
        float sc[FP];
        float mx = -1e30f;
#pragma unroll
        for (int kf = 0; kf < FP; ++kf) {
            sc[kf] = -1e30f;
            if (kf < F) {
                const h8 k0 = *(const h8*)&kv_s[wave][0][kf][sl * 16], k1 = *(const h8*)&kv_s[wave][0][kf][sl * 16 + 8];
                float d = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) d += (float)q0[j] * (float)k0[j] + (float)q1[j] * (float)k1[j];
                d += __shfl_xor(d, 16, 64);
                d += __shfl_xor(d, 32, 64);
                sc[kf] = d * c;
                mx = fmaxf(mx, sc[kf]);
            }
        }
        float l = 0.f;
#pragma unroll
        for (int kf = 0; kf < FP; ++kf) {
            sc[kf] = (kf < F) ? __builtin_amdgcn_exp2f(sc[kf] - mx) : 0.f;
            l += sc[kf];
        }
        const float inv = 1.f / l;
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
        for (int kf = 0; kf < FP; ++kf) {
            if (kf < F) {
                const h8 v0 = *(const h8*)&kv_s[wave][1][kf][sl * 16], v1 = *(const h8*)&kv_s[wave][1][kf][sl * 16 + 8];
                const float w = sc[kf] * inv;
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc[j] += w * (float)v0[j]; acc[8 + j] += w * (float)v1[j]; }
            }
        }
        if (qok) {
            half_t* op = (half_t*)a.O + ((size_t)(b * Fq + fq) * a.HW + p) * a.ldo + col;
            h8 o0, o1;
#pragma unroll
            for (int j = 0; j < 8; ++j) { o0[j] = (half_t)acc[j]; o1[j] = (half_t)acc[8 + j]; }
            *(h8*)op = o0;
            *(h8*)(op + 8) = o1;
        }
    }
}

}  // namespace

// attention_d64.hip: the head_dim-64 long-sequence kernel family (variant 0 = the kernel above)
bool flash_attn_d64_applies(const AttnArgs& a);
int op_flash_attn_d64(const AttnArgs& a, hipStream_t s, int variant);
int attn_variant();

static int flash_attn_launch(const AttnArgs& a, hipStream_t s);
int op_flash_attn(const AttnArgs& a, hipStream_t s) {
    if (t_collect) {                                 // lock-step replay of sibling blocks: deposited, launched by the collector's flush()
        int rc = 0;
        const int i = t_collect->slot(OpCollector::ATTN, s, &rc);
        if (i < 0) return rc;
        t_collect->at[i] = a;
        return 0;
    }
    return flash_attn_launch(a, s);
}
int op_flash_attn_group(const AttnArgs* a, int n, hipStream_t s) {
    CTRL_CHECK(a && n >= 1 && n <= kMaxGroup, "flash_attn_group: 1..4 problems");
    auto pk = [](const void* p) { return (int)((uintptr_t)p & 15); };
    bool same = n > 1 && group_launches_enabled();
    for (int i = 1; same && i < n; ++i)
        same = a[i].ldq == a[0].ldq && a[i].ldk == a[0].ldk && a[i].Lkpad == a[0].Lkpad && a[i].kvB == a[0].kvB && a[i].ldo == a[0].ldo &&
               a[i].B == a[0].B && a[i].heads == a[0].heads && a[i].D == a[0].D && a[i].Lq == a[0].Lq && a[i].Lk == a[0].Lk &&
               a[i].scale == a[0].scale && a[i].k_prescaled == a[0].k_prescaled && pk(a[i].Q) == pk(a[0].Q) && pk(a[i].K) == pk(a[0].K) &&
               pk(a[i].Vt) == pk(a[0].Vt) && pk(a[i].O) == pk(a[0].O);
    if (!same) {
        for (int i = 0; i < n; ++i) TRY(flash_attn_launch(a[i], s));
        return 0;
    }
    t_attn_grp = a; t_attn_grp_n = n;        // the launch functions pick the siblings up from here
    const int rc = flash_attn_launch(a[0], s);
    t_attn_grp = nullptr; t_attn_grp_n = 1;
    return rc;
}
static int flash_attn_launch(const AttnArgs& a, hipStream_t s) {
    for (int i = 0; i < (t_attn_grp ? t_attn_grp_n : 1); ++i) {
        const AttnArgs& q = t_attn_grp ? t_attn_grp[i] : a;
        CTRL_CHECK((((uintptr_t)q.Q | (uintptr_t)q.K | (uintptr_t)q.Vt) & 15) == 0 && ((uintptr_t)q.O & 7) == 0,
                   "flash_attn: pointers must be 16-byte aligned");
    }
    CTRL_CHECK(a.B > 0 && a.heads > 0 && a.Lq > 0 && a.Lk > 0, "flash_attn: empty problem");
    CTRL_CHECK(a.kvB == 1 || a.kvB == a.B, "flash_attn: kvB must be 1 or B");
    CTRL_CHECK(a.Lkpad % 64 == 0 && a.Lkpad >= a.Lk, "flash_attn: Lkpad must be a multiple of 64 and >= Lk");
    CTRL_CHECK(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldo % 4 == 0, "flash_attn: leading dims must be multiples of 8");
    CTRL_CHECK((((uintptr_t)a.Q | (uintptr_t)a.K | (uintptr_t)a.Vt) & 15) == 0 && ((uintptr_t)a.O & 7) == 0,
               "flash_attn: pointers must be 16-byte aligned");
    switch (a.D) {
        case 64:
            if (attn_variant() > 0 && flash_attn_d64_applies(a)) return op_flash_attn_d64(a, s, attn_variant());
            // long sequences: 256 queries per workgroup halve the K/V LDS-DMA traffic per FLOP (the limiter at L = 16384)
            if (a.Lq >= 2048 && !policy_raw(P_ATTN_NW4)) return launch_attn<64, 8, false>(a, s);
            return launch_attn<64, 4, true>(a, s);
        case 40: return launch_attn<40, 4, true>(a, s);
        case 80: return launch_attn<80, 4, true>(a, s);
        case 160: return launch_attn<160, 4, true>(a, s);
        default: CTRL_FAIL("flash_attn: unsupported head_dim " + std::to_string(a.D) + " (supported: 40, 64, 80, 160)");
    }
}

int op_temporal_attn(const TAttnArgs& a_in, hipStream_t s) {
    TAttnArgs a = a_in;
    // descriptor defaults of the unsharded form: queries = keys = all F frames, K|V in the same rows as Q
    if (a.Fq <= 0) a.Fq = a.F;
    if (a.Fl <= 0) a.Fl = a.F;
    if (!a.KV) { a.KV = (const half_t*)a.Q + (size_t)a.heads * 64; a.ldkv = a.ld; }
    CTRL_CHECK(a.F > 0 && a.F <= 32 && a.Fq <= a.F && a.F % a.Fl == 0, "temporal_attn: F must be in 1..32 (Fq <= F, Fl | F)");
    CTRL_CHECK(a.ld % 8 == 0 && a.ldo % 8 == 0 && a.ldkv % 8 == 0, "temporal_attn: leading dims must be multiples of 8");
    CTRL_CHECK((((uintptr_t)a.Q | (uintptr_t)a.KV | (uintptr_t)a.O) & 15) == 0, "temporal_attn: pointers must be 16-byte aligned");
    const long nitems = (long)a.Bc * a.HW * a.heads;
    CTRL_CHECK(nitems > 0, "temporal_attn: empty problem");
    dim3 grid((unsigned)((nitems + 3) / 4));
    PROF_WORK(4.0 * nitems * a.Fq * a.F * 64, 2.0 * nitems * 64 * (2.0 * a.Fq + 2.0 * a.F));
    prof_detail("clips%d Fq%d F%d HW%d heads%d", a.Bc, a.Fq, a.F, a.HW, a.heads);
    if (a.F <= 16) LAUNCH("temporal_attn", temporal_attn_kernel<16>, grid, dim3(256), 0, s, a);
    else LAUNCH("temporal_attn", temporal_attn_kernel<32>, grid, dim3(256), 0, s, a);
    return 0;
}
