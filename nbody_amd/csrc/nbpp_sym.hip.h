// nbpp_sym.hip.h - direct summation that evaluates every unordered pair once.
//
// The reference evaluates each ordered pair separately ("Newton's third law not exploited",
// its i-loop owns whole particles: src/particles.cpp:270-365).  Here one evaluation of
// s = (1/q)^(3/2) serves both directions: a_i += G m_j s d and a_j -= G m_i s d, which is the
// same pair arithmetic (src/particles.cpp:349-359) applied to (i,j) and to (j,i), where the
// reference's two displacements are exact negatives of each other.
//
// Work decomposition.  The sorted particle array is cut into tiles of 64 and groups of I tiles.
// A wavefront owns one group (lane l holds I particles of it in registers) and visits the tiles
// of the next half of all groups in circular order, so every unordered pair of groups is
// visited by exactly one wavefront and all wavefronts do equal work.  A visited tile lives in
// registers, one particle per lane, together with the three reaction sums of that particle;
// after every step the whole record is rotated by one lane (DPP wave_ror:1), so that after 64
// steps every particle of the group has met every particle of the tile and the records are back
// in their home lanes.  No LDS, no barriers.  Reaction sums and the group's own sums are added
// to the output with fp64 global atomics (other wavefronts add to the same particles), so the
// result is not bitwise reproducible from run to run; the ordered kernel k_pairs is.
//
// The minimum image is decided per (group, tile) exactly as in k_pairs: uniform shift (applied
// to the visiting tile's private coordinates), single-axis straddle, or per-pair wrap.
#pragma once
#include "nbpp_kernels.hip.h"

namespace nbpp {

struct SymArgs {
    const double* sx; const double* sy; const double* sz; const double* sgm;
    const int* sbox;        // [n_tiles][8]
    const int* order;       // sorted -> original index
    int n;                  // real particles (sorted entries beyond n are zero-weight padding)
    int n_tiles;            // padded count / 64, a multiple of I
    // The visit list of every group is cut into n_slices equal slices; this launch handles
    // slices [slice_begin, slice_begin + gridDim.y) (blockIdx.y picks one).  Slices of one launch
    // fill the GPU when there are few groups; slices of different launches are the shares of a
    // multi-GPU run.  Slice 0 also does the group's own tiles.
    int slice_begin;
    int n_slices;
    double* ax; double* ay; double* az;   // length n, original order, accumulated atomically
    const int* uniform_mass;
    const double* common_gm;
    unsigned long long* tile_counts;
    PairConsts k;
};

__device__ __forceinline__ double rotate_lane(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // wave_ror:1, every lane enabled: the "old" operand is never kept, so it is tied to the
    // source to spare the compiler a zero-initialising move per rotation
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x13C, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x13C, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// (Eight tiles per group instead of four would halve the rotations per pair again, but needs
// more than 168 registers: at 3 waves/SIMD with spills it ran at 2.27 instead of 2.89 T
// interactions/s; six fit 168 registers without spills in the loop and ran at 2.77 T/s at
// 3 waves/SIMD: the wavefront lost costs more than the rotations saved.)
// (Moving the coordinates through the LDS crossbar instead, ds_bpermute_b32 issued at the start
// of a step, was measured slower: the extra live registers spill at 4 waves/SIMD and 3 waves/SIMD
// loses more than the six DPP moves cost.)
// (Nor does keeping the visited tile in LDS help - every lane reading entry (lane + step) mod 64 and
// adding its reaction there with ds_add_f64, no rotation at all, ten VALU instructions fewer per
// step of 104: 1.67 s against 1.63 s per 128^3 evaluation, 1.74 s with the next record requested a
// step ahead, which spills.)

// One evaluation, both directions.  (dx,dy,dz) points from i to j.
template <bool SOFT0, bool WEIGHTED>
__device__ __forceinline__ void pair_both(double dx, double dy, double dz, double wi, double wj,
                                          const PairConsts& k, double& ax, double& ay, double& az,
                                          double& rx, double& ry, double& rz) {
    const double q = __builtin_fma(dx, dx, __builtin_fma(dy, dy, __builtin_fma(dz, dz, k.eps2)));
    const double y = __builtin_amdgcn_rsq(q);
    const double y2 = y * y;
    const double h = __builtin_fma(-q, y2, 1.0);
    const double y3 = y * y2;
    const double c = h * __builtin_fma(k.c1875, h, k.c15);
    double s = __builtin_fma(y3, c, y3);
    if constexpr (SOFT0) s = (q > 0.0) ? s : 0.0;
    const double fi = WEIGHTED ? wj * s : s;   // acts on i, weight of j
    const double fj = WEIGHTED ? wi * s : s;   // acts on j, weight of i
    ax = __builtin_fma(fi, dx, ax);
    ay = __builtin_fma(fi, dy, ay);
    az = __builtin_fma(fi, dz, az);
    rx = __builtin_fma(-fj, dx, rx);
    ry = __builtin_fma(-fj, dy, ry);
    rz = __builtin_fma(-fj, dz, rz);
}

// KIND 0: uniform image shift already applied to the tile; 1..3: axis KIND-1 straddles +-L/2
// (see straddle_tile); 4: per-pair wrap on all axes.  REACT = false is the diagonal case (a tile
// of the group itself): ordered pairs, no reaction sums.
template <int I, int KIND, bool SOFT0, bool WEIGHTED, bool REACT>
__device__ __forceinline__ void visit_tile(const PairConsts& k, double sigma, double xj, double yj,
                                           double zj, double wj, const double (&xi)[I],
                                           const double (&yi)[I], const double (&zi)[I],
                                           const double (&wi)[I], double (&ax)[I], double (&ay)[I],
                                           double (&az)[I], double& rx, double& ry, double& rz) {
    constexpr int AXIS = KIND - 1;
    const double nsig = -sigma, half = k.half_L;
    if constexpr (KIND >= 1 && KIND <= 3) {  // the straddling axis is accumulated sigma-flipped
#pragma unroll
        for (int u = 0; u < I; ++u) {
            if (AXIS == 0) ax[u] *= sigma;
            if (AXIS == 1) ay[u] *= sigma;
            if (AXIS == 2) az[u] *= sigma;
        }
    }
#pragma unroll 1
    for (int step = 0; step < 64; ++step) {
#pragma unroll
        for (int u = 0; u < I; ++u) {
            double d[3] = {xj - xi[u], yj - yi[u], zj - zi[u]};
            if constexpr (KIND >= 1 && KIND <= 3) {
                const double e = __builtin_fma(nsig, d[AXIS], half);
                d[AXIS] = __builtin_copysign(half, e) - e;
            } else if constexpr (KIND == 4) {
                d[0] = min_image(d[0], k.L, half);
                d[1] = min_image(d[1], k.L, half);
                d[2] = min_image(d[2], k.L, half);
            }
            if constexpr (REACT) {
                pair_both<SOFT0, WEIGHTED>(d[0], d[1], d[2], wi[u], wj, k, ax[u], ay[u], az[u], rx, ry, rz);
            } else {
                pair_direct<SOFT0, WEIGHTED>(d[0], d[1], d[2], wj, k, ax[u], ay[u], az[u]);
            }
        }
        xj = rotate_lane(xj); yj = rotate_lane(yj); zj = rotate_lane(zj);
        if constexpr (WEIGHTED) wj = rotate_lane(wj);
        if constexpr (REACT) { rx = rotate_lane(rx); ry = rotate_lane(ry); rz = rotate_lane(rz); }
    }
    if constexpr (KIND >= 1 && KIND <= 3) {
#pragma unroll
        for (int u = 0; u < I; ++u) {
            if (AXIS == 0) ax[u] *= sigma;
            if (AXIS == 1) ay[u] *= sigma;
            if (AXIS == 2) az[u] *= sigma;
        }
        if constexpr (REACT) {
            if (AXIS == 0) rx *= sigma;
            if (AXIS == 1) ry *= sigma;
            if (AXIS == 2) rz *= sigma;
        }
    }
}

template <int I, bool SOFT0, bool WEIGHTED, bool REACT>
__device__ __forceinline__ void visit_dispatch(int kind, const PairConsts& k, double sigma, double xj,
                                               double yj, double zj, double wj, const double (&xi)[I],
                                               const double (&yi)[I], const double (&zi)[I],
                                               const double (&wi)[I], double (&ax)[I], double (&ay)[I],
                                               double (&az)[I], double& rx, double& ry, double& rz) {
    switch (kind) {
        case 0: visit_tile<I, 0, SOFT0, WEIGHTED, REACT>(k, sigma, xj, yj, zj, wj, xi, yi, zi, wi, ax, ay, az, rx, ry, rz); break;
        case 1: visit_tile<I, 1, SOFT0, WEIGHTED, REACT>(k, sigma, xj, yj, zj, wj, xi, yi, zi, wi, ax, ay, az, rx, ry, rz); break;
        case 2: visit_tile<I, 2, SOFT0, WEIGHTED, REACT>(k, sigma, xj, yj, zj, wj, xi, yi, zi, wi, ax, ay, az, rx, ry, rz); break;
        case 3: visit_tile<I, 3, SOFT0, WEIGHTED, REACT>(k, sigma, xj, yj, zj, wj, xi, yi, zi, wi, ax, ay, az, rx, ry, rz); break;
        default: visit_tile<I, 4, SOFT0, WEIGHTED, REACT>(k, sigma, xj, yj, zj, wj, xi, yi, zi, wi, ax, ay, az, rx, ry, rz); break;
    }
}

constexpr int HOME_FLUSH = 64;   // visits between two flushes of a group's own sums (power of 2)

template <int I, bool SOFT0>
__global__ void __launch_bounds__(WG, 4) k_pairs_sym(const SymArgs a) {
    const int lane = threadIdx.x & 63;
    const PairConsts k = a.k;
    const int T = a.n_tiles, NG = T / I;
    const int group = __builtin_amdgcn_readfirstlane(blockIdx.x * (WG / 64) + (threadIdx.x >> 6));
    if (group >= NG) return;   // whole wavefronts leave together; nothing is shared
    const int slice = a.slice_begin + blockIdx.y;
    const bool uniform = __builtin_amdgcn_readfirstlane(a.uniform_mass[0]) != 0;

    double xi[I], yi[I], zi[I], wi[I], ax[I], ay[I], az[I];
    int ilo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, ihi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
#pragma unroll
    for (int u = 0; u < I; ++u) {
        const int p = (group * I + u) * 64 + lane;
        xi[u] = a.sx[p]; yi[u] = a.sy[p]; zi[u] = a.sz[p]; wi[u] = a.sgm[p];
        ax[u] = 0.0; ay[u] = 0.0; az[u] = 0.0;
        const int* b = a.sbox + (group * I + u) * 8;
#pragma unroll
        for (int c = 0; c < 3; ++c) { ilo[c] = min(ilo[c], b[2 * c]); ihi[c] = max(ihi[c], b[2 * c + 1]); }
    }
    // padding (weight 0) fills the sorted array from index n up to a multiple of 256: every
    // tile from n/64 on may hold some.  Anything touching such a tile takes the weighted form,
    // everything else the equal-mass form when masses are equal.
    const int first_pad_tile = a.n / 64;
    const bool group_has_pad = (group * I + I - 1 >= first_pad_tile);
    unsigned n_visits = 0, n_slow = 0;

    // ---- the group against its own tiles: ordered pairs, no reaction ----
#pragma unroll 1
    for (int u2 = 0; u2 < (slice == 0 ? I : 0); ++u2) {
        const int p = (group * I + u2) * 64 + lane;
        double rx = 0, ry = 0, rz = 0;
        // a group is compact, but a tiny box can still put members half a box apart: wrap per pair
        visit_dispatch<I, SOFT0, true, false>(4, k, 1.0, a.sx[p], a.sy[p], a.sz[p], a.sgm[p], xi, yi, zi, wi,
                                              ax, ay, az, rx, ry, rz);
    }

    // ---- the next half of all groups, circularly ----
    const int half_groups = (NG - 1) / 2;
    const int n_groups = half_groups + ((NG % 2 == 0 && group < NG / 2) ? 1 : 0);
    const long long n_visits_all = static_cast<long long>(n_groups) * I;
    const int visit_begin = static_cast<int>(n_visits_all * slice / a.n_slices);
    const int visit_end = static_cast<int>(n_visits_all * (slice + 1) / a.n_slices);
    int b = static_cast<int>((static_cast<long long>((group + 1) % NG) * I + visit_begin) % T);
    const double scale = uniform ? a.common_gm[0] : 1.0;
#pragma unroll 1
    for (int visit = visit_begin; visit < visit_end; ++visit) {
        const int p = b * 64 + lane;
        double xj = a.sx[p], yj = a.sy[p], zj = a.sz[p], wj = a.sgm[p];
        double rx = 0.0, ry = 0.0, rz = 0.0;
        const int* box = a.sbox + b * 8;
        int kind = 0, n_straddle = 0, n_generic = 0;
        double sigma = 1.0;
        double shift[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int dlo = box[2 * c] - ihi[c], dhi = box[2 * c + 1] - ilo[c];
            if (dhi <= QHALF - 1 && dlo >= -(QHALF - 1)) {
            } else if (dlo >= QHALF + 1) {
                shift[c] = k.L;
            } else if (dhi <= -(QHALF + 1)) {
                shift[c] = -k.L;
            } else if (dlo >= QHALF / 2) {
                ++n_straddle; kind = c + 1; sigma = 1.0;
            } else if (dhi <= -(QHALF / 2)) {
                ++n_straddle; kind = c + 1; sigma = -1.0;
            } else {
                ++n_generic;
            }
        }
        if (n_generic > 0 || n_straddle > 1) kind = 4;
        if (kind != 4) {   // uniform shifts go onto the tile's private coordinates
            if (kind != 1) xj -= shift[0];
            if (kind != 2) yj -= shift[1];
            if (kind != 3) zj -= shift[2];
        }
        ++n_visits;
        if (kind != 0) ++n_slow;
        const bool weighted = !uniform || group_has_pad || b >= first_pad_tile;
        if (weighted)
            visit_dispatch<I, SOFT0, true, true>(kind, k, sigma, xj, yj, zj, wj, xi, yi, zi, wi, ax, ay, az, rx, ry, rz);
        else
            visit_dispatch<I, SOFT0, false, true>(kind, k, sigma, xj, yj, zj, wj, xi, yi, zi, wi, ax, ay, az, rx, ry, rz);
        if (p < a.n) {
            const int o = a.order[p];
            unsafeAtomicAdd(&a.ax[o], rx * scale);
            unsafeAtomicAdd(&a.ay[o], ry * scale);
            unsafeAtomicAdd(&a.az[o], rz * scale);
        }
        if (++b == T) b = 0;
        // The group's own sums go out every HOME_FLUSH visits, not once at the end: a register
        // sum that runs over millions of terms rounds at the magnitude of its largest partial
        // sum (the pull of half the box) every time, and short chains keep that error an order
        // of magnitude below that of one long sequential sum, for 3*I atomics per 64 visits.
        if (((visit - visit_begin) & (HOME_FLUSH - 1)) == HOME_FLUSH - 1) {
#pragma unroll
            for (int u = 0; u < I; ++u) {
                const int ph = (group * I + u) * 64 + lane;
                if (ph < a.n) {
                    const int o = a.order[ph];
                    unsafeAtomicAdd(&a.ax[o], ax[u] * scale);
                    unsafeAtomicAdd(&a.ay[o], ay[u] * scale);
                    unsafeAtomicAdd(&a.az[o], az[u] * scale);
                }
                ax[u] = 0.0; ay[u] = 0.0; az[u] = 0.0;
            }
        }
    }

#pragma unroll
    for (int u = 0; u < I; ++u) {
        const int p = (group * I + u) * 64 + lane;
        if (p < a.n) {
            const int o = a.order[p];
            unsafeAtomicAdd(&a.ax[o], ax[u] * scale);
            unsafeAtomicAdd(&a.ay[o], ay[u] * scale);
            unsafeAtomicAdd(&a.az[o], az[u] * scale);
        }
    }
    if (a.tile_counts && lane == 0) {
        atomicAdd(&a.tile_counts[0], static_cast<unsigned long long>(n_visits));
        atomicAdd(&a.tile_counts[1], static_cast<unsigned long long>(n_slow));
    }
}

}  // namespace nbpp
