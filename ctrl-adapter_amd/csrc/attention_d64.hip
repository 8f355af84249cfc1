// Flash attention, head_dim 64, long sequences (the adapter's spatial self-attention at 128x128 / 64x64 latents:
// model/adapter_spatial_temporal.py:108-116 called :271 -> diffusers Attention / F.scaled_dot_product_attention;
// 16384 / 4096 tokens, heads = C/64).  Same algorithm and data flow as flash_attn_kernel<64, 8, false, true> in
// attention.hip (swapped QK^T so a lane owns one query, P fed to P.V straight from registers, K pre-scaled by
// softmax_scale*log2(e), running maximum folded into the QK^T accumulator, 3-deep LDS-DMA ring of 64-key K / V^T tiles) --
// rebuilt around what the round-2 counters showed: that loop issues ~170 VALU instructions against 16 MFMAs per 64-key
// tile and wave (VALU pipe ~1.6x the matrix pipe), 47 of them integer address arithmetic for the 16 fragment reads.
//   * the ring is unrolled by its depth, so the slot of a tile is a compile-time constant that lands in the
//     ds_read_b128 offset field; the XOR swizzle of a fragment address is `base ^ (k-step << 5)` of ONE per-lane base
//     (rows r and r + 32 share their swizzle and differ by an immediate): 8 loop-invariant address registers, no
//     integer VALU in the loop;
//   * the -m accumulator initialisation is a 16-register tuple handed to the first MFMA of a block as its C operand,
//     rewritten only when the running maximum moves (no v_mov per tile);
//   * the row-max exchange between the two half-waves is v_permlane32_swap (VALU) instead of ds_bpermute (an LDS
//     round trip that also waits for the fragment reads in flight);
//     (row sums of P through v_dot2c_f32_f16 on the packed probabilities were tried as well: no faster, dropped);
//   * QB = 2: a wave owns two 32-query blocks, every K / V^T fragment read feeds two MFMAs (half the LDS read traffic
//     per FLOP -- at one block per wave the LDS pipe is as busy as the matrix pipe) and the two blocks' independent
//     softmax / MFMA chains give the scheduler work to overlap inside one wave.
// Requirements (checked by the dispatcher): D == 64, K pre-scaled, Lk % 64 == 0, kvB == B.
#include "ops.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

enum { AO_NEGM = 1, AO_SGB = 2, AO_PRIO = 4, AO_DEFER = 8, AO_MINI = 16, AO_DEFER8 = 32, AO_VPRIO = 64 };

template <int QB, int NW, int OPT, int NSTAGE = 3>
__global__ __launch_bounds__(NW * 64, (QB == 1 ? 4 : 2)) void flash_attn_d64_kernel(AttnGroup kargs, int per, const half_t* zeros) {
    const int gp = __builtin_amdgcn_readfirstlane(blockIdx.x / per);      // grouped launch: attention.hip, AttnGroup
    const int gbid = blockIdx.x - gp * per;
    const AttnArgs a = ATTN_GROUP_ARGS(gp);
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "ring depth 2 .. 4");
    constexpr int TILEB = 64 * 64 * 2;                 // bytes of one K (or V^T) tile
    constexpr int VOFF = NSTAGE * TILEB;               // V^T ring behind the K ring
    constexpr int PASSES = 512 / (NW * 64);            // 16-byte chunks of a tile per lane
    constexpr int LPT = 2 * PASSES;
    constexpr int QPW = 32 * QB;                       // queries per wave
    static_assert(PASSES >= 1, "workgroup too large for the tile");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform: LDS-DMA destinations and per-wave offsets stay in SGPRs)
    const int hi = lane >> 5, lq = lane & 31;
    // XCD-aware work map, see flash_attn_kernel: every XCD owns whole (batch, head) pairs
    const int qtiles = (a.Lq + NW * QPW - 1) / (NW * QPW);
    int pair, qt;
    if (!attn_work_map(gbid, qtiles, a.B * a.heads, &pair, &qt)) return;
    const int b = pair / a.heads, h = pair - b * a.heads;
    const int q0 = qt * (NW * QPW) + wave * QPW;
    const int Lq = a.Lq, Lk = a.Lk;

    // ---- Q fragments (B operand: column = query, k = hi*8 + j) ----
    h8 qf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        int q = q0 + qb * 32 + lq;
        if (q > Lq - 1) q = Lq - 1;
        const half_t* qp = (const half_t*)a.Q + ((size_t)b * Lq + q) * a.ldq + (size_t)h * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *(const h8*)(qp + ks * 16 + hi * 8);
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[qb][ks]));      // retire the loads before the loop (see attention.hip)

    const half_t* Kbase = (const half_t*)a.K + (size_t)b * Lk * a.ldk + (size_t)h * 64;
    const half_t* Vbase = (const half_t*)a.Vt + ((size_t)b * a.heads + h) * 64 * (size_t)a.Lkpad;

    // ---- staging: chunk ci of a tile -> K row ci/8, 16-byte piece (ci%8) ^ swizzle; V^T row ci/8 likewise ----
    unsigned k_off[PASSES], v_off[PASSES];      // bytes from the (batch, head) base: < 4 GiB by the size of the operands
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        const int ci = (i * NW + wave) * 64 + lane;
        const int r = ci >> 3, p = ci & 7;
        const int src = p ^ ((r >> 1) & 7);
        k_off[i] = (unsigned)(((size_t)r * a.ldk + src * 8) * sizeof(half_t));
        v_off[i] = (unsigned)(((size_t)r * a.Lkpad + src * 8) * sizeof(half_t));
    }
    typedef const void __attribute__((address_space(1)))* gptr_t;
    typedef void __attribute__((address_space(3)))* lptr_t;
    auto stage = [&](int t, int slot) {
        const int kt0 = t * 64;
#pragma unroll
        for (int i = 0; i < PASSES; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)((const char*)(Kbase + (size_t)kt0 * a.ldk) + k_off[i]),
                                             (lptr_t)(smem_raw + slot * TILEB + (i * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < PASSES; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)((const char*)(Vbase + kt0) + v_off[i]),
                                             (lptr_t)(smem_raw + VOFF + slot * TILEB + (i * NW + wave) * 1024), 16, 0, 0);
    };
    (void)zeros;

    // ---- fragment read addresses (bytes inside a tile), loop invariant ----
    // K row of lane: krow (bits 2 <-> 3 of lq swapped, so that the lane's scores are its own P^T B-operand); rows krow
    // and krow + 32 (the two 32-key blocks) share the swizzle (r >> 1) & 7 and differ by 4096 bytes.
    const int krow = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    const int kA = krow * 128 + ((hi ^ ((krow >> 1) & 7)) << 4);
    const int vA = lq * 128 + ((hi ^ ((lq >> 1) & 7)) << 4);

    f16v o[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][db][r] = 0.f;
    float m_run[QB], l_run[QB];
    f16v negm[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = 0.f; l_run[qb] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[qb][r] = 0.f;
    }

    const int ntiles = Lk >> 6;
#pragma unroll
    for (int t = 0; t < NSTAGE - 1; ++t)
        if (t < ntiles) stage(t, t);


    auto tile = [&](auto slot_c, const int t) {
        constexpr int SLOT = decltype(slot_c)::value;
        if (t + NSTAGE - 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // Every fragment read of tile t-1 is RETIRED before this barrier (round 6): the workgroup re-stages that tile's ring slot right
        // behind it, and hipcc, left alone, schedules the barrier ABOVE the tile's last MFMA and the lgkmcnt wait of its operand -- the
        // read is then still in flight when another wave's LDS-DMA overwrites the slot.  Found as a 3-8 % per-replay flake of the captured,
        // multi-lane adapter forward (one wave's 32 queries of one head slightly off; tools/diag/graph_replay_stress.py); never seen with
        // one kernel at a time on the chip.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // tile t visible to every wave; the slot of tile t-1 is free
        if (t + NSTAGE - 1 < ntiles) stage(t + NSTAGE - 1, (SLOT + NSTAGE - 1) % NSTAGE);
        const char* Kb = smem_raw + SLOT * TILEB;
        const char* Vb = smem_raw + VOFF + SLOT * TILEB;

        // One softmax step covers KB 32-key blocks: 2 = the whole tile (scores of 64 keys live at once), 1 (AO_MINI) = a
        // block at a time -- 16 fewer live score registers, which is what lets the -m tuple stay resident at 128
        // registers per wave (2 workgroups of 8 waves per CU); the price is a second max exchange per tile.
        constexpr int KB = (OPT & AO_MINI) ? 1 : 2;
#pragma unroll
        for (int m0 = 0; m0 < 2; m0 += KB) {
            // ---- S^T = K . Q^T - m ----
            f16v sacc[QB][KB];
            if constexpr (OPT & AO_PRIO) __builtin_amdgcn_s_setprio(1);      // matrix clusters win the issue arbitration
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int mi = 0; mi < KB; ++mi) {
                    const h8 kf = *(const h8*)(Kb + (kA ^ (ks << 5)) + (m0 + mi) * 4096);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        if (ks == 0) {
                            if constexpr (OPT & AO_NEGM) {
                                sacc[qb][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qb][0], negm[qb], 0, 0, 0);
                            } else {
                                f16v c0;
#pragma unroll
                                for (int r = 0; r < 16; ++r) c0[r] = -m_run[qb];
                                sacc[qb][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qb][0], c0, 0, 0, 0);
                            }
                        } else {
                            sacc[qb][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qb][ks], sacc[qb][mi], 0, 0, 0);
                        }
                    }
                }
            if constexpr (OPT & AO_PRIO) __builtin_amdgcn_s_setprio(0);
            if constexpr (OPT & AO_VPRIO) __builtin_amdgcn_s_setprio(1);     // the softmax section wins the VALU arbitration
            // register r of block mb holds key  kt0 + 32*mb + 16*(r>>3) + 8*hi + (r&7)
            // ---- online softmax (exp2 domain; sacc = s - m_run already) ----
            h8 pf[QB][KB][2];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float mx = sacc[qb][0][0];
#pragma unroll
                for (int mi = 0; mi < KB; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = vmaxf(mx, sacc[qb][mi][r]);
                {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                    mx = vmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                }
                // mx > 0 <=> the row maximum moved (first step: m_run = 0 stands for "none yet" and the update is forced).
                // AO_DEFER: the reference point is only moved when a score exceeds it by more than 2^4 -- p <= 16 is as exact
                // in fp16 / fp32 as p <= 1, and a slowly creeping maximum no longer costs a rescale per tile
                constexpr float THR = (OPT & AO_DEFER8) ? 8.f : ((OPT & AO_DEFER) ? 4.f : 0.f);
                const bool first = (t == 0 && m0 == 0);
                if (first || __any(mx > THR)) {        // wave-uniform
                    const float d = first ? mx : fmaxf(mx, 0.f);
                    const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-d);
                    l_run[qb] *= alpha;
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;
                    m_run[qb] += d;
#pragma unroll
                    for (int mi = 0; mi < KB; ++mi)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sacc[qb][mi][r] -= d;
                    if constexpr (OPT & AO_NEGM) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) negm[qb][r] = -m_run[qb];
                    }
                }
                typedef unsigned u4v __attribute__((ext_vector_type(4)));
                float psum = 0.f;
#pragma unroll
                for (int mi = 0; mi < KB; ++mi)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        u4v w;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int r = s2 * 8 + i * 2;
                            const float p0 = __builtin_amdgcn_exp2f(sacc[qb][mi][r]), p1 = __builtin_amdgcn_exp2f(sacc[qb][mi][r + 1]);
                            const h2 pk = {(half_t)p0, (half_t)p1};
                            w[i] = __builtin_bit_cast(unsigned, pk);
                            psum += p0 + p1;
                        }
                        pf[qb][mi][s2] = __builtin_bit_cast(h8, w);
                    }
                l_run[qb] += psum;
            }

            // ---- O^T += V^T . P^T ----
            if constexpr (OPT & AO_VPRIO) __builtin_amdgcn_s_setprio(0);
            if constexpr (OPT & AO_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mi = 0; mi < KB; ++mi)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const h8 vf = *(const h8*)(Vb + (vA ^ (((m0 + mi) * 2 + s2) << 5)) + db * 4096);
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb)
                            o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[qb][mi][s2], o[qb][db], 0, 0, 0);
                    }
            if constexpr (OPT & AO_PRIO) __builtin_amdgcn_s_setprio(0);
        }
    };

    int t = 0;
    if constexpr (NSTAGE == 3) {
        for (; t + 3 <= ntiles; t += 3) {
            tile(std::integral_constant<int, 0>{}, t);
            tile(std::integral_constant<int, 1>{}, t + 1);
            tile(std::integral_constant<int, 2>{}, t + 2);
        }
        if (t < ntiles) { tile(std::integral_constant<int, 0>{}, t); ++t; }
        if (t < ntiles) { tile(std::integral_constant<int, 1>{}, t); ++t; }
    } else if constexpr (NSTAGE == 4) {
        for (; t + 4 <= ntiles; t += 4) {
            tile(std::integral_constant<int, 0>{}, t);
            tile(std::integral_constant<int, 1>{}, t + 1);
            tile(std::integral_constant<int, 2>{}, t + 2);
            tile(std::integral_constant<int, 3>{}, t + 3);
        }
        if (t < ntiles) { tile(std::integral_constant<int, 0>{}, t); ++t; }
        if (t < ntiles) { tile(std::integral_constant<int, 1>{}, t); ++t; }
        if (t < ntiles) { tile(std::integral_constant<int, 2>{}, t); ++t; }
    } else {
        for (; t + 2 <= ntiles; t += 2) {
            tile(std::integral_constant<int, 0>{}, t);
            tile(std::integral_constant<int, 1>{}, t + 1);
        }
        if (t < ntiles) { tile(std::integral_constant<int, 0>{}, t); ++t; }
    }

    // ---- finalize: O[q][d] = O^T[d][q] / l ----
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = 1.0f / l_tot;
        const int q = q0 + qb * 32 + lq;
        if (q < Lq) {
            half_t* op = (half_t*)a.O + ((size_t)b * Lq + q) * a.ldo + (size_t)h * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // rows (r&3) + 8*(r>>2) + 4*hi of the 32x32 C/D fragment: r = 4g..4g+3 -> 4 consecutive d
                    const int d0 = db * 32 + 8 * g + 4 * hi;
                    h4 v = {(half_t)(o[qb][db][4 * g] * inv), (half_t)(o[qb][db][4 * g + 1] * inv),
                            (half_t)(o[qb][db][4 * g + 2] * inv), (half_t)(o[qb][db][4 * g + 3] * inv)};
                    *(h4*)(op + d0) = v;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Software-pipelined form: a wave owns TWO 32-query blocks A and B that run half a tile apart, so that inside ONE wave's
// instruction stream the matrix work of one block always has the other block's softmax beside it:
//     phase 1 of tile t:   MFMA  P.V of B (tile t-1), then QK^T of B (tile t)      VALU  exp2 / pack / row sums of A (tile t)
//     phase 2 of tile t:   MFMA  P.V of A (tile t),   then QK^T of A (tile t+1)    VALU  exp2 / pack / row sums of B (tile t)
// (between the phases: the row-maximum reduction of the block whose scores just finished and -- rarely -- its rescale).
// Each phase is one basic block of 16 MFMAs (512 matrix-pipe cycles) and ~100 independent VALU instructions, which the
// scheduler can interleave; the round-2 kernel left this overlap to chance (waves of a workgroup are phase-locked by the
// per-tile barrier: its counters showed 38 % of the wave cycles waiting behind the matrix pipe and 36 % at waits).
// 4 waves x 64 queries per workgroup, two workgroups per CU (<= 256 registers); K and V^T rings of 3 tiles each: phase 2 of
// tile t needs V(t) and K(t+1), which are waited for once per tile, before ONE barrier.
template <int OPT>
__global__ __launch_bounds__(256, 2) void flash_attn_d64p_kernel(AttnGroup kargs, int per) {
    const int gp = __builtin_amdgcn_readfirstlane(blockIdx.x / per);
    const int gbid = blockIdx.x - gp * per;
    const AttnArgs a = ATTN_GROUP_ARGS(gp);
    constexpr int NW = 4, NSTAGE = 3;
    constexpr int TILEB = 64 * 64 * 2;
    constexpr int VOFF = NSTAGE * TILEB;
    constexpr int PASSES = 512 / (NW * 64);            // 2
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform: LDS-DMA destinations and per-wave offsets stay in SGPRs)
    const int hi = lane >> 5, lq = lane & 31;
    const int qtiles = (a.Lq + 255) / 256;
    int pair, qt;
    if (!attn_work_map(gbid, qtiles, a.B * a.heads, &pair, &qt)) return;
    const int b = pair / a.heads, h = pair - b * a.heads;
    const int q0 = qt * 256 + wave * 64;
    const int Lq = a.Lq, Lk = a.Lk;

    h8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int q = q0 + qb * 32 + lq;
        if (q > Lq - 1) q = Lq - 1;
        const half_t* qp = (const half_t*)a.Q + ((size_t)b * Lq + q) * a.ldq + (size_t)h * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *(const h8*)(qp + ks * 16 + hi * 8);
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[qb][ks]));

    const half_t* Kbase = (const half_t*)a.K + (size_t)b * Lk * a.ldk + (size_t)h * 64;
    const half_t* Vbase = (const half_t*)a.Vt + ((size_t)b * a.heads + h) * 64 * (size_t)a.Lkpad;
    unsigned k_off[PASSES], v_off[PASSES];
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        const int ci = (i * NW + wave) * 64 + lane;
        const int r = ci >> 3, p = ci & 7;
        const int src = p ^ ((r >> 1) & 7);
        k_off[i] = (unsigned)(((size_t)r * a.ldk + src * 8) * sizeof(half_t));
        v_off[i] = (unsigned)(((size_t)r * a.Lkpad + src * 8) * sizeof(half_t));
    }
    typedef const void __attribute__((address_space(1)))* gptr_t;
    typedef void __attribute__((address_space(3)))* lptr_t;
    auto stage_k = [&](int t, int slot) {
#pragma unroll
        for (int i = 0; i < PASSES; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)((const char*)(Kbase + (size_t)t * 64 * a.ldk) + k_off[i]),
                                             (lptr_t)(smem_raw + slot * TILEB + (i * NW + wave) * 1024), 16, 0, 0);
    };
    auto stage_v = [&](int t, int slot) {
#pragma unroll
        for (int i = 0; i < PASSES; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)((const char*)(Vbase + t * 64) + v_off[i]),
                                             (lptr_t)(smem_raw + VOFF + slot * TILEB + (i * NW + wave) * 1024), 16, 0, 0);
    };

    const int krow = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    const int kA = krow * 128 + ((hi ^ ((krow >> 1) & 7)) << 4);
    const int vA = lq * 128 + ((hi ^ ((lq >> 1) & 7)) << 4);

    f16v o[2][2], sc[2][2];
    h8 pf[2][2][2];
    float m_run[2], l_run[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        m_run[qb] = 0.f; l_run[qb] = 0.f;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[qb][x][r] = 0.f; sc[qb][x][r] = 0.f; }
#pragma unroll
            for (int y = 0; y < 2; ++y) pf[qb][x][y] = h8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    const int ntiles = Lk >> 6;
    // the first P.V of block B multiplies an all-zero P with V^T slot 2: that slot must hold finite values
    {
        const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = tid; i < TILEB / 16; i += 256) *(h8*)(smem_raw + VOFF + 2 * TILEB + i * 16) = z;
    }
    stage_k(0, 0);
    if (ntiles > 1) stage_k(1, 1);
    stage_v(0, 0);

    // QK^T of block qb against the K tile at byte offset kb of LDS; acc starts at -m (scores arrive referenced to it)
    auto qk = [&](const int qb, const int kb) {
        f16v c0;
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[r] = -m_run[qb];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const h8 kf = *(const h8*)(smem_raw + kb + (kA ^ (ks << 5)) + mb * 4096);
                sc[qb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qb][ks], ks == 0 ? c0 : sc[qb][mb], 0, 0, 0);
            }
    };
    auto pv = [&](const int qb, const int vb) {
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const h8 vf = *(const h8*)(smem_raw + VOFF + vb + (vA ^ (c4 << 5)) + db * 4096);
                o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[qb][c4 >> 1][c4 & 1], o[qb][db], 0, 0, 0);
            }
    };
    // row maximum of the finished scores of block qb; moves the reference point (rarely) -- ends a basic block
    auto smax = [&](const int qb, const bool first) {
        float mx = sc[qb][0][0];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = vmaxf(mx, sc[qb][mb][r]);
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = vmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        constexpr float THR = (OPT & AO_DEFER) ? 4.f : 0.f;
        if (first || __any(mx > THR)) {
            const float d = first ? mx : fmaxf(mx, 0.f);
            const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-d);
            l_run[qb] *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;
            m_run[qb] += d;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[qb][mb][r] -= d;
        }
    };
    // exp2 / pack / row sum of block qb (straight-line: shares a basic block with the other block's MFMAs)
    auto sexp = [&](const int qb) {
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        // pin the start of the softmax arithmetic HERE (pure VALU code is otherwise free to be emitted ahead of the phase's
        // scheduling region, where it runs before the MFMAs instead of beside them)
        asm volatile("" : "+v"(sc[qb][0]), "+v"(sc[qb][1]));
        float psum = 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u4v w;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = s2 * 8 + i * 2;
                    const float p0 = __builtin_amdgcn_exp2f(sc[qb][mb][r]), p1 = __builtin_amdgcn_exp2f(sc[qb][mb][r + 1]);
                    const h2 pk = {(half_t)p0, (half_t)p1};
                    w[i] = __builtin_bit_cast(unsigned, pk);
                    psum += p0 + p1;
                }
                pf[qb][mb][s2] = __builtin_bit_cast(h8, w);
            }
        l_run[qb] += psum;
    };
    // scheduling hint for one phase: 16 x { 1 LDS read, 1 MFMA, 2 exp2, 4 VALU } (the tail takes whatever is left)
    auto hint = [&]() {
        if constexpr (OPT & AO_SGB) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one fragment read
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);     // two exp2 (transcendentals are not in the VALU class)
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);     // pack / row sum / accumulator init
            }
        }
    };

    // AO_MINI: a phase cut into four pinned quarters -- {4 MFMAs of block qm, exp2 / pack / sum of 8 scores of block qs} each,
    // every quarter its own scheduling region (sched_barrier on both sides, the scores it exponentiates made opaque at its
    // start so the arithmetic cannot be emitted ahead of it), with a {1 read, 1 MFMA, 2 exp2, 3 VALU} x 4 hint inside
    auto phase_q = [&](const int qm, const int qs, const int vb, const int kb) {
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        f16v c0;
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[r] = -m_run[qm];
        float psum = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int mb = q >> 1, s2 = q & 1;
            // Ordering point.  MFMA and exp2 are pure operations: nothing orders them against a sched_barrier or a volatile asm
            // unless their operands pass through it -- so the point takes (whole register tuples, in place) the scores the next
            // two quarters exponentiate, the accumulators the previous quarter's MFMAs wrote, the row sum and the packed
            // probabilities (the previous quarter's VALU results): that quarter is complete in program order above this line.
            if (q == 0) asm volatile("" : "+v"(sc[qs][0]));
            else if (q == 1) asm volatile("" : "+v"(o[qm][0]), "+v"(o[qm][1]), "+v"(psum), "+v"(pf[qs][0][0]));
            else if (q == 2) asm volatile("" : "+v"(sc[qs][1]), "+v"(o[qm][0]), "+v"(o[qm][1]), "+v"(psum), "+v"(pf[qs][0][1]));
            else asm volatile("" : "+v"(sc[qm][0]), "+v"(sc[qm][1]), "+v"(psum), "+v"(pf[qs][1][0]));
            float x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = sc[qs][mb][s2 * 8 + i];
            // ---- 4 MFMAs: quarters 0,1 = P.V (c4 = 2q, 2q+1; both d blocks), quarters 2,3 = QK^T (ks = 2(q-2), +1; both key blocks)
            if (q < 2) {
#pragma unroll
                for (int c4 = 2 * q; c4 < 2 * q + 2; ++c4)
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const h8 vf = *(const h8*)(smem_raw + VOFF + vb + (vA ^ (c4 << 5)) + db * 4096);
                        o[qm][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[qm][c4 >> 1][c4 & 1], o[qm][db], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int ks = 2 * (q - 2); ks < 2 * (q - 2) + 2; ++ks)
#pragma unroll
                    for (int mb2 = 0; mb2 < 2; ++mb2) {
                        const h8 kf = *(const h8*)(smem_raw + kb + (kA ^ (ks << 5)) + mb2 * 4096);
                        sc[qm][mb2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qm][ks], ks == 0 ? c0 : sc[qm][mb2], 0, 0, 0);
                    }
            }
            // ---- softmax of 8 scores of the other block
            u4v w;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p0 = __builtin_amdgcn_exp2f(x[2 * i]), p1 = __builtin_amdgcn_exp2f(x[2 * i + 1]);
                const h2 pk = {(half_t)p0, (half_t)p1};
                w[i] = __builtin_bit_cast(unsigned, pk);
                psum += p0 + p1;
            }
            pf[qs][mb][s2] = __builtin_bit_cast(h8, w);
            if constexpr (OPT & AO_SGB) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" : "+v"(sc[qm][0]), "+v"(sc[qm][1]), "+v"(psum), "+v"(pf[qs][1][1]));
        l_run[qs] += psum;
    };

    // prologue: scores of block A for tile 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    qk(0, 0);
    smax(0, true);

    auto step = [&](auto slot_c, const int t) {
        constexpr int S0 = decltype(slot_c)::value, S1 = (S0 + 1) % 3, S2 = (S0 + 2) % 3;     // t % 3, (t+1) % 3, (t+2) % 3 = (t-1) % 3
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // K(t+1), V(t) (issued one tile ago) have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and this wave's reads of the slots re-staged below are retired (see flash_attn_d64_kernel)
        __builtin_amdgcn_s_barrier();
        if (t + 2 < ntiles) stage_k(t + 2, S2);
        if (t + 1 < ntiles) stage_v(t + 1, S1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (OPT & AO_MINI) {
            phase_q(1, 0, S2 * TILEB, S0 * TILEB);
            smax(1, t == 0);
            __builtin_amdgcn_sched_barrier(0);
            phase_q(0, 1, S0 * TILEB, S1 * TILEB);
            if (t + 1 < ntiles) smax(0, false);
        } else {
            // ---- phase 1 ----
            pv(1, S2 * TILEB);                  // B, tile t-1 (all-zero P on the first tile)
            qk(1, S0 * TILEB);                  // B, tile t
            sexp(0);                            // A, tile t
            hint();
            __builtin_amdgcn_sched_barrier(0);
            smax(1, t == 0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- phase 2 ----
            pv(0, S0 * TILEB);                  // A, tile t
            qk(0, S1 * TILEB);                  // A, tile t+1 (past the last tile: stale finite data, result unused)
            sexp(1);                            // B, tile t
            hint();
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < ntiles) smax(0, false);
        }
    };
    int t = 0;
    for (; t + 3 <= ntiles; t += 3) {
        step(std::integral_constant<int, 0>{}, t);
        step(std::integral_constant<int, 1>{}, t + 1);
        step(std::integral_constant<int, 2>{}, t + 2);
    }
    if (t < ntiles) { step(std::integral_constant<int, 0>{}, t); ++t; }
    if (t < ntiles) { step(std::integral_constant<int, 1>{}, t); ++t; }
    // epilogue: P.V of block B for the last tile (runtime slot)
    pv(1, ((ntiles - 1) % 3) * TILEB);

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = 1.0f / l_tot;
        const int q = q0 + qb * 32 + lq;
        if (q < Lq) {
            half_t* op = (half_t*)a.O + ((size_t)b * Lq + q) * a.ldo + (size_t)h * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = db * 32 + 8 * g + 4 * hi;
                    h4 v = {(half_t)(o[qb][db][4 * g] * inv), (half_t)(o[qb][db][4 * g + 1] * inv),
                            (half_t)(o[qb][db][4 * g + 2] * inv), (half_t)(o[qb][db][4 * g + 3] * inv)};
                    *(h4*)(op + d0) = v;
                }
        }
    }
}

template <int OPT>
int launch_d64p(const AttnArgs& a, hipStream_t s) {
    constexpr size_t smem = (size_t)3 * 2 * 64 * 64 * sizeof(half_t);
    static bool attr_done[kMaxDevices] = {};
    const int dev = cur_device();
    if (!attr_done[dev]) {
        HIP_TRY(hipFuncSetAttribute((const void*)flash_attn_d64p_kernel<OPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_done[dev] = true;
    }
    const int qtiles = (a.Lq + 255) / 256, pairs = a.B * a.heads;
    const int per = 8 * ((pairs + 7) / 8) * qtiles, G = attn_grp_count();
    dim3 grid((unsigned)(per * G));
    PROF_WORK(G * 4.0 * a.B * a.heads * (double)a.Lq * a.Lk * a.D, G * 2.0 * a.heads * a.D * (2.0 * a.B * a.Lq + 2.0 * a.kvB * a.Lk));
    if (G > 1) prof_detail("B%d h%d D%d Lq%d Lk%d x%d", a.B, a.heads, a.D, a.Lq, a.Lk, G);
    else prof_detail("B%d h%d D%d Lq%d Lk%d", a.B, a.heads, a.D, a.Lq, a.Lk);
    prof_symbol("flash_attn_d64p_kernel<%d>", OPT);
    LAUNCH("flash_attn", (flash_attn_d64p_kernel<OPT>), grid, dim3(256), smem, s, attn_grp_make(a), per);
    return 0;
}

template <int QB, int NW, int OPT, int NSTAGE = 3>
int launch_d64(const AttnArgs& a, hipStream_t s) {
    constexpr size_t smem = (size_t)NSTAGE * 2 * 64 * 64 * sizeof(half_t);
    static bool attr_done[kMaxDevices] = {};
    const int dev = cur_device();
    if (!attr_done[dev]) {
        HIP_TRY(hipFuncSetAttribute((const void*)flash_attn_d64_kernel<QB, NW, OPT, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_done[dev] = true;
    }
    const half_t* zeros = (const half_t*)device_zero_page();
    const int qtiles = (a.Lq + NW * 32 * QB - 1) / (NW * 32 * QB), pairs = a.B * a.heads;
    const int per = 8 * ((pairs + 7) / 8) * qtiles, G = attn_grp_count();
    dim3 grid((unsigned)(per * G));
    PROF_WORK(G * 4.0 * a.B * a.heads * (double)a.Lq * a.Lk * a.D, G * 2.0 * a.heads * a.D * (2.0 * a.B * a.Lq + 2.0 * a.kvB * a.Lk));
    if (G > 1) prof_detail("B%d h%d D%d Lq%d Lk%d x%d", a.B, a.heads, a.D, a.Lq, a.Lk, G);
    else prof_detail("B%d h%d D%d Lq%d Lk%d", a.B, a.heads, a.D, a.Lq, a.Lk);
    prof_symbol("flash_attn_d64_kernel<%d, %d, %d, %d>", QB, NW, OPT, NSTAGE);
    LAUNCH("flash_attn", (flash_attn_d64_kernel<QB, NW, OPT, NSTAGE>), grid, dim3(NW * 64), smem, s, attn_grp_make(a), per, zeros);
    return 0;
}

int g_variant = -1;

}  // namespace

bool flash_attn_d64_applies(const AttnArgs& a) {
    return a.D == 64 && a.k_prescaled && a.Lk % 64 == 0 && a.Lk >= 128 && a.kvB == a.B && a.Lq >= 2048;
}

// variant: performance only (every variant computes the same function; they differ in instruction selection and in
// how many queries a wave owns).  0 = the round-2 kernel in attention.hip; default = CTRL_ATTN_VARIANT or the best measured.
int attn_set_variant(int v) { g_variant = v; return 0; }
int attn_variant() {
    if (g_variant < 0) {
        const char* e = policy_raw(P_ATTN_VARIANT);
        g_variant = e ? atoi(e) : 2;
        if (g_variant < 0 || g_variant > 14) {        // (a bad value used to surface as "unknown variant" inside every long-sequence forward)
            fprintf(stderr, "ctrl: CTRL_ATTN_VARIANT=%s is not a variant (0..14), using the default\n", e);
            g_variant = 2;
        }
    }
    return g_variant;
}

int op_flash_attn_d64(const AttnArgs& a, hipStream_t s, int variant) {
    switch (variant) {
        case 1: return launch_d64<1, 8, 0>(a, s);
        case 2: return launch_d64<1, 8, AO_DEFER>(a, s);
        case 3: return launch_d64<1, 8, AO_DEFER8>(a, s);
        case 4: return launch_d64<1, 8, AO_DEFER | AO_PRIO>(a, s);
        case 5: return launch_d64<1, 8, AO_DEFER, 2>(a, s);
        case 6: return launch_d64<1, 4, AO_DEFER, 2>(a, s);
        case 7: return launch_d64<2, 4, AO_DEFER>(a, s);
        case 8: return launch_d64<2, 4, AO_DEFER | AO_PRIO>(a, s);
        case 9: return launch_d64p<AO_DEFER>(a, s);
        case 10: return launch_d64p<AO_MINI | AO_DEFER>(a, s);
        case 11: return launch_d64<1, 8, AO_MINI | AO_NEGM | AO_DEFER>(a, s);
        case 12: return launch_d64<1, 8, AO_DEFER, 4>(a, s);
        case 13: return launch_d64<2, 8, AO_DEFER>(a, s);
        case 14: return launch_d64<1, 8, AO_DEFER | AO_VPRIO>(a, s);
        default: CTRL_FAIL("flash_attn: unknown variant " + std::to_string(variant));
    }
}
