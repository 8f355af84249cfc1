// Walk order of the implicit-GEMM output tiles (igemm.hip): which (tile_m, tile_n) a workgroup computes.
//
// Workgroups are dispatched round-robin over the 8 XCDs (workgroup b runs on XCD b % 8, the k = b / 8-th one there) and each
// XCD has its own 4 MiB L2, so the order decides what the ≈32-64 workgroups resident on one XCD at a time have in common:
//
//   legacy   every XCD walks a contiguous range of the M-major tile list (all N tiles of an activation panel, then the next
//            panel).  Activations are fetched once; the XCD streams the WHOLE weight matrix once per ≈2 panels.  Fine while the
//            weights fit beside the streams in L2; measured 15x the algorithmic fetch for the 512 -> 4096 GEGLU GEMM at
//            M = 131072 (4 MiB of weights: profiles/r02_pmc_hbm_traffic_v2.json, 2.09 GB fetched against 0.14 GB).
//   xcd_m    the XCD owns ntm / 8 activation panels and walks them once per GROUP of G weight panels (group-major, then M,
//            then N inside the group): the G weight panels stay L2-resident for the whole sweep and every activation panel is
//            fetched ceil(ntn / G) times.  Needs ntm % 8 == 0.
//   xcd_n    the XCD owns ntn / 8 weight panels (a slice of the weight matrix nobody else reads) and every activation panel,
//            same group walk inside: for small-M, wide-N problems (mid-level GEGLU, 2048 x 10240 x 1280) where eight copies
//            of the activations are cheaper than eight copies of the weights.  Needs ntn % 8 == 0.
//
// Every mode is a bijection of [0, ntm * ntn) (tests/test_host.py walks them through ctrl_igemm_tile_of); a mode whose
// divisibility condition does not hold falls back to legacy.
#pragma once
#include <hip/hip_runtime.h>

namespace tileorder {

enum { ORDER_LEGACY = 0, ORDER_XCD_M = 1, ORDER_XCD_N = 2 };

// order word: bits 16.. = mode, bits 0..15 = group width G in tiles (0 = the whole width)
__host__ __device__ inline int make_order(int mode, int group) { return (mode << 16) | (group & 0xffff); }

// k-th tile of an mt x nt rectangle walked in N-groups of G tiles (the last group may be narrower)
__host__ __device__ inline void walk_rect(int k, int mt, int nt, int G, int* m, int* n) {
    if (G <= 0 || G >= nt) {
        *m = k / nt;
        *n = k - *m * nt;
        return;
    }
    const int full = nt / G, per = mt * G, g = k / per;
    if (g < full) {
        const int kk = k - g * per;
        *m = kk / G;
        *n = g * G + (kk - *m * G);
    } else {
        const int rem = nt - full * G, kk = k - full * per;
        *m = kk / rem;
        *n = full * G + (kk - *m * rem);
    }
}

__host__ __device__ inline void tile_of(int bid, int ntm, int ntn, int order, int* tile_m, int* tile_n) {
    const int mode = order >> 16, G = order & 0xffff;
    const int xcd = bid & 7, k = bid >> 3;
    if (mode == ORDER_XCD_M && (ntm & 7) == 0) {
        int m, n;
        walk_rect(k, ntm >> 3, ntn, G, &m, &n);
        *tile_m = xcd * (ntm >> 3) + m;
        *tile_n = n;
        return;
    }
    if (mode == ORDER_XCD_N && (ntn & 7) == 0) {
        int m, n;
        walk_rect(k, ntm, ntn >> 3, G, &m, &n);
        *tile_m = m;
        *tile_n = xcd * (ntn >> 3) + n;
        return;
    }
    // legacy: XCD x walks the x-th contiguous range of the M-major list (ranges differ by at most one tile)
    const int nblk = ntm * ntn, q = nblk >> 3, r = nblk & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    *tile_m = t / ntn;
    *tile_n = t - *tile_m * ntn;
}

}  // namespace tileorder
