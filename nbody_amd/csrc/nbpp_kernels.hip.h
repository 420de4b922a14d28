// nbpp_kernels.hip.h - gfx950 device code of libnbpp: binning, the pair kernels, and the
// bandwidth-bound caller-contract kernels.  Included once, by nbpp_api.hip.
//
// Data layout in HBM: every particle array is SoA fp64 (x[], y[], z[], ...), coalesced
// by lane.  The pair kernels read a library-owned, spatially sorted copy of the sources
// (sx, sy, sz, sgm = G*m) and of the targets (tx, ty, tz, tperm).
//
// Pair arithmetic follows reference src/particles.cpp:321-359 (identical at :212-254):
// minimum-image displacement, r^2, optional cutoff / cubic taper / subtractive
// mesh-softened term, Plummer softening.  Algebraically the reference's
//     a = G m / q ;  a * d / sqrt(q)         (q = r^2 + eps^2)
// is evaluated here as d * (G m) * q^(-3/2) with q^(-3/2) from the 24-bit v_rsq_f64 seed and
// one third-order correction (relative error ~3e-16 per pair).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <stdint.h>

namespace nbpp {

constexpr int WG = 256;        // threads per workgroup = 4 wavefronts of 64
constexpr int TJ = 256;        // sources staged in LDS per tile (8 KiB of 32-B records)
constexpr int SUB = 64;        // sources per image-classified sub-tile
constexpr int QBITS = 20;      // bounding boxes are kept in units of L / 2^20
constexpr int QHALF = 1 << (QBITS - 1);

// ---------------------------------------------------------------------------------
// Scalars of one call.  Mirrors nbpp_params plus derived reciprocals.
// ---------------------------------------------------------------------------------
struct PairConsts {
    double L, half_L;       // domain size, L/2            (src/math_utils.h:12)
    double eps2;            // SOFTENING_SQUARED           (src/particles.cpp:145)
    double eps_pm2;         // (0.5*CELL_SIZE)^2           (src/particles.cpp:146-147)
    double r_switch;        // R_SWITCH_START              (:148)
    double r_switch_sq;     // (:149)
    double cutoff_sq;       // (:150)
    double inv_width;       // 1 / CUTOFF_TRANSITION_WIDTH (:151, :333)
    double qscale;          // 2^QBITS / L
    double c15, c1875;      // 1.5 and 1.875, passed as data so that they live in registers
                            // (as literals the compiler re-materialises them per pair)
};

// ---------------------------------------------------------------------------------
// q^(-3/2) from a 24-bit seed y = v_rsq_f64(q): h = 1 - q y^2, then
//   q^(-3/2) = y^3 (1 - h)^(-3/2) = y^3 (1 + h (3/2 + 15/8 h) + O(h^3)),  |h^3| < 2^-69.
// ---------------------------------------------------------------------------------
// With a zero guard, for SOFTENING_SQUARED == 0, where q can be 0 for the self pair or for
// coincident particles.
// The reference skips i == j by index (src/particles.cpp:202,302) and produces NaN for
// two distinct coincident particles with zero softening; this returns 0 for both.
__device__ __forceinline__ double inv_cube_root2_guarded(double q) {
    const double y = __builtin_amdgcn_rsq(q);
    const double y2 = y * y;
    const double h = __builtin_fma(-q, y2, 1.0);
    const double y3 = y * y2;
    const double c = h * __builtin_fma(1.875, h, 1.5);
    const double r = __builtin_fma(y3, c, y3);
    return (q > 0.0) ? r : 0.0;
}

// r = q^(-1/2) to fp64 accuracy from the same seed: y (1 + h (1/2 + 3/8 h)).
__device__ __forceinline__ double inv_sqrt_refined(double q) {
    const double y = __builtin_amdgcn_rsq(q);
    const double h = __builtin_fma(-q * y, y, 1.0);
    const double c = h * __builtin_fma(0.375, h, 0.5);
    return __builtin_fma(y, c, y);
}

// src/math_utils.h:11-16, the same two strict compares.
__device__ __forceinline__ double min_image(double d, double L, double half_L) {
    d = (d > half_L) ? d - L : d;
    d = (d < -half_L) ? d + L : d;
    return d;
}

// One source acting on one target, direct summation (use_pm == 0).
//   (dx,dy,dz) minimum-image displacement, gm = G*m_j.          src/particles.cpp:349-359
// 16 fp64 VALU instructions + one v_rsq_f64 per pair:
//   q = eps^2 + dz^2 + dy^2 + dx^2 (3 fma) ; y = rsq(q) ; y2 = y*y ; h = 1 - q*y2 ;
//   w = (gm*y)*y2 ; s = w + w*h*(1.5 + 1.875 h) ; a += s*d (3 fma)        [+ 3 sub for d]
// v_rsq_f64 has a relative error below 2^-24, so |h| < 2^-23 and the dropped h^3 term of
// (1-h)^(-3/2) is below 2^-67.
// WEIGHTED = false is the equal-mass form: the common factor G*m is applied once to the
// finished sums, which removes one multiply per pair (15 instead of 16).
template <bool SOFT0, bool WEIGHTED = true>
__device__ __forceinline__ void pair_direct(double dx, double dy, double dz, double gm,
                                            const PairConsts& k, double& ax, double& ay,
                                            double& az) {
    const double q = __builtin_fma(dx, dx, __builtin_fma(dy, dy, __builtin_fma(dz, dz, k.eps2)));
    const double y = __builtin_amdgcn_rsq(q);
    const double y2 = y * y;
    const double h = __builtin_fma(-q, y2, 1.0);
    const double w = WEIGHTED ? (gm * y) * y2 : y * y2;
    const double c = h * __builtin_fma(k.c1875, h, k.c15);
    double s = __builtin_fma(w, c, w);
    if constexpr (SOFT0) s = (q > 0.0) ? s : 0.0;  // self pair / coincident particles
    ax = __builtin_fma(s, dx, ax);
    ay = __builtin_fma(s, dy, ay);
    az = __builtin_fma(s, dz, az);
}

// One source acting on one target, P3M short range (use_pm != 0).  src/particles.cpp:324-359
//   r^2 > cutoff_sq                -> nothing               (:326-328)
//   r^2 > r_switch_sq              -> S = 2x^3 - 3x^2 + 1   (:330-335)
//   a += S * (pp - pm_short) * d                            (:339-359)
__device__ __forceinline__ void pair_p3m(double dx, double dy, double dz, double gm,
                                         const PairConsts& k, double& ax, double& ay,
                                         double& az) {
    const double r2 = __builtin_fma(dx, dx, __builtin_fma(dy, dy, dz * dz));
    if (r2 > k.cutoff_sq) return;
    double f = inv_cube_root2_guarded(r2 + k.eps2) - inv_cube_root2_guarded(r2 + k.eps_pm2);
    if (r2 > k.r_switch_sq) {
        const double dist = r2 * inv_sqrt_refined(r2);
        const double t = (dist - k.r_switch) * k.inv_width;
        const double S = __builtin_fma(t * t, __builtin_fma(2.0, t, -3.0), 1.0);
        f *= S;
    }
    const double s = gm * f;
    ax = __builtin_fma(s, dx, ax);
    ay = __builtin_fma(s, dy, ay);
    az = __builtin_fma(s, dz, az);
}

// ---------------------------------------------------------------------------------
// Binning: counting sort of particles by cell.  Replaces the serial head/next linked
// list of reference src/particles.cpp:50,60-62,81-83,94-96 (src/particles.h:8-24).
// Final order is (cell, original index): deterministic from run to run.
// ---------------------------------------------------------------------------------
struct BinGrid {
    double scale;   // cells per unit length
    int nside;      // cells per axis
    int morton;     // 1: Z-order cell ids (nside must be a power of two <= 1024)
};

// Cell of a coordinate in [0, L); anything outside (or a product that rounds up to nside)
// is clamped to the nearest edge cell.
__device__ __forceinline__ int bin_coord(double x, double scale, int nside) {
    double c = floor(x * scale);
    c = fmin(fmax(c, 0.0), static_cast<double>(nside - 1));
    return static_cast<int>(c);
}

__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__device__ __forceinline__ int cell_id(double x, double y, double z, const BinGrid g) {
    const int ix = bin_coord(x, g.scale, g.nside);
    const int iy = bin_coord(y, g.scale, g.nside);
    const int iz = bin_coord(z, g.scale, g.nside);
    if (g.morton) return static_cast<int>(spread3(ix) | (spread3(iy) << 1) | (spread3(iz) << 2));
    return (iz * g.nside + iy) * g.nside + ix;  // x fastest, as src/particles.cpp:94
}

__global__ void k_bin_count(int n, const double* __restrict__ x, const double* __restrict__ y,
                            const double* __restrict__ z, BinGrid g, int* __restrict__ cell,
                            int* __restrict__ rank, int* __restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cell_id(x[i], y[i], z[i], g);
    cell[i] = c;
    rank[i] = atomicAdd(&count[c], 1);
}

// Exclusive scan in three launches (block sums, scan of the sums, scan with the block's offset):
// the input is read twice and the output written once, all in 16-byte accesses in which
// neighbouring lanes touch neighbouring addresses.  A block covers SCAN_SUBS sub-blocks of 1024
// entries; both arrays must be 16-byte aligned.
constexpr int SCAN_SUB = WG * 4;
constexpr int SCAN_SUBS = 4;
constexpr int SCAN_BLOCK = SCAN_SUB * SCAN_SUBS;

__device__ __forceinline__ int block_exclusive_scan(int v, int* lds, int& total) {
    // inclusive scan over 256 threads via wave shuffles + 4 wave totals in LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += lds[w];
    total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + inc - v;
}

__device__ __forceinline__ int4 scan_load(const int* __restrict__ in, int i, int n) {   // zeros past n
    if (i + 3 < n) return *reinterpret_cast<const int4*>(in + i);
    return make_int4(i < n ? in[i] : 0, i + 1 < n ? in[i + 1] : 0, i + 2 < n ? in[i + 2] : 0, 0);
}

__global__ void __launch_bounds__(WG) k_scan_reduce(int n, const int* __restrict__ in,
                                                    int* __restrict__ block_sum) {
    __shared__ int lds[WG / 64];
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_SUBS; ++k) {
        const int4 v = scan_load(in, blockIdx.x * SCAN_BLOCK + k * SCAN_SUB + threadIdx.x * 4, n);
        s += (v.x + v.y) + (v.z + v.w);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// One workgroup scans the per-block totals in place (exclusive) and stores the grand total
// at out_total[0].
__global__ void __launch_bounds__(WG) k_scan_totals(int nblocks, int* __restrict__ block_sum,
                                                    int* __restrict__ out_total) {
    __shared__ int lds[4];
    int carry = 0;
    for (int start = 0; start < nblocks; start += WG) {
        const int i = start + threadIdx.x;
        const int v = (i < nblocks) ? block_sum[i] : 0;
        int total;
        const int ex = block_exclusive_scan(v, lds, total);
        if (i < nblocks) block_sum[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) out_total[0] = carry;
}

// out[i] = block offset + sum of the block's entries before i
__global__ void __launch_bounds__(WG) k_scan_apply(int n, const int* __restrict__ in,
                                                   int* __restrict__ out,
                                                   const int* __restrict__ block_sum) {
    __shared__ int lds[WG / 64];
    int carry = block_sum[blockIdx.x];
#pragma unroll 1
    for (int k = 0; k < SCAN_SUBS; ++k) {
        const int i = blockIdx.x * SCAN_BLOCK + k * SCAN_SUB + threadIdx.x * 4;
        const int4 v = scan_load(in, i, n);
        int total;
        const int run = carry + block_exclusive_scan((v.x + v.y) + (v.z + v.w), lds, total);
        const int4 o = make_int4(run, run + v.x, run + v.x + v.y, run + v.x + v.y + v.z);
        if (i + 3 < n) {
            *reinterpret_cast<int4*>(out + i) = o;
        } else {
            if (i < n) out[i] = o.x;
            if (i + 1 < n) out[i + 1] = o.y;
            if (i + 2 < n) out[i + 2] = o.z;
        }
        carry += total;
    }
}

__global__ void k_bin_scatter(int n, const int* __restrict__ cell, const int* __restrict__ rank,
                              const int* __restrict__ start, int* __restrict__ order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    order[start[cell[i]] + rank[i]] = i;
}

// The atomic ranks above depend on arrival order; putting each cell's members in
// ascending original index makes the sorted order a pure function of the input.
// Three sizes of cell.  Up to CANON_TINY members: the one thread that looks at the cell sorts it
// by insertion.  Up to 64: the cell goes on the `mid` list and one wavefront sorts it by rank in
// registers (k_cell_canon_mid).  More (the core of a collapsed halo holds thousands): the `big`
// list and a whole workgroup (k_cell_canon_big).  Insertion by one thread needs O(members^2)
// dependent steps while the rest of the GPU waits: in a clustered 128^3 set with the limit at 48
// members this kernel alone took as long as all the other binning passes together.
constexpr int CANON_TINY = 8;
constexpr int CANON_WAVE = 64;

__global__ void k_cell_canon(int ncell, const int* __restrict__ start, int* __restrict__ order,
                             int* __restrict__ mid, int* __restrict__ big) {   // list[0] = count, list[1..] = cell ids
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = c < ncell;
    const int b = live ? start[c] : 0, e = live ? start[c + 1] : 0;
    if (e - b > CANON_WAVE) big[1 + atomicAdd(&big[0], 1)] = c;   // few
    // slots on the mid list: one atomic per wavefront, not per cell (tens of thousands of cells
    // would otherwise queue up on one counter)
    const bool to_mid = e - b > CANON_TINY && e - b <= CANON_WAVE;
    const unsigned long long votes = __ballot(to_mid);
    if (votes) {
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll(static_cast<long long>(votes)) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&mid[0], __popcll(votes));
        base = __shfl(base, leader, 64);
        if (to_mid) mid[1 + base + __popcll(votes & ((1ull << lane) - 1ull))] = c;
    }
    if (e - b > CANON_TINY) return;
    for (int p = b + 1; p < e; ++p) {
        const int v = order[p];
        int q = p - 1;
        while (q >= b && order[q] > v) {
            order[q + 1] = order[q];
            --q;
        }
        order[q + 1] = v;
    }
}

// One wavefront per listed cell (9 to 64 members): every lane holds one member, its rank is the
// number of members with a smaller entry (entries are distinct), counted by passing every member
// through all lanes.
__global__ void __launch_bounds__(WG) k_cell_canon_mid(const int* __restrict__ start, int* __restrict__ order,
                                                       const int* __restrict__ mid) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * WG + threadIdx.x) >> 6, n_waves = (gridDim.x * WG) >> 6;
    const int n_mid = mid[0];
    for (int k = wave; k < n_mid; k += n_waves) {
        const int c = mid[1 + k];
        const int b = start[c], cnt = start[c + 1] - b;      // wave-uniform
        const int v = lane < cnt ? order[b + lane] : 0x7fffffff;
        int rank = 0;
        for (int u = 0; u < cnt; ++u) rank += __shfl(v, u, 64) < v ? 1 : 0;
        if (lane < cnt) order[b + rank] = v;                  // every lane read its member before any lane writes
    }
}

// The listed cells, one workgroup per cell at a time.  Up to CANON_LDS members: bitonic sort in
// LDS (a cell of 5000 members takes ~1500 compare-exchanges per thread; counting ranks, the
// method below, would take 98000 comparisons per thread, half a millisecond for that one cell).
// Beyond that: rank sort staged through LDS - the rank of a member is the number of members with
// a smaller entry (entries are distinct); tmp_val / tmp_rank: scratch as long as `order`.
constexpr int CANON_LDS = 8192;

__global__ void __launch_bounds__(WG) k_cell_canon_big(const int* __restrict__ start, int* __restrict__ order,
                                                       const int* __restrict__ big, int* __restrict__ tmp_val,
                                                       int* __restrict__ tmp_rank) {
    __shared__ int tile[CANON_LDS];
    const int n_big = big[0];
    for (int k = blockIdx.x; k < n_big; k += gridDim.x) {
        const int c = big[1 + k];
        const int b = start[c], e = start[c + 1];
        const int cnt = e - b;                                  // the same for every thread: barriers line up
        if (cnt <= CANON_LDS) {
            int n2 = 128;
            while (n2 < cnt) n2 <<= 1;
            __syncthreads();
            for (int u = threadIdx.x; u < n2; u += WG) tile[u] = u < cnt ? order[b + u] : 0x7fffffff;
            for (int span = 2; span <= n2; span <<= 1) {
                for (int j = span >> 1; j > 0; j >>= 1) {
                    __syncthreads();
                    for (int t = threadIdx.x; t < (n2 >> 1); t += WG) {
                        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                        const int a = tile[lo], z = tile[hi];
                        const bool ascending = (lo & span) == 0;
                        if ((a > z) == ascending) { tile[lo] = z; tile[hi] = a; }
                    }
                }
            }
            __syncthreads();
            for (int u = threadIdx.x; u < cnt; u += WG) order[b + u] = tile[u];
            continue;
        }
        const int rounds = (cnt + WG - 1) / WG;
        for (int r = 0; r < rounds; ++r) {
            const int p = b + r * WG + threadIdx.x;
            const int v = p < e ? order[p] : 0;
            int rank = 0;
            for (int t0 = b; t0 < e; t0 += CANON_LDS) {     // members staged through LDS
                const int len = min(CANON_LDS, e - t0);
                __syncthreads();
                for (int u = threadIdx.x; u < len; u += WG) tile[u] = order[t0 + u];
                __syncthreads();
                if (p < e)
                    for (int u = 0; u < len; ++u) rank += tile[u] < v ? 1 : 0;
            }
            if (p < e) { tmp_val[p] = v; tmp_rank[p] = rank; }
        }
        __syncthreads();
        for (int p = b + threadIdx.x; p < e; p += WG) order[b + tmp_rank[p]] = tmp_val[p];
        __syncthreads();
    }
}

// Gather the sorted copy.  Entries [n, npad) replicate the last particle with zero mass,
// so a padded tile never widens a bounding box and never contributes.
// `uniform` (may be null) points at the equal-mass flag: then the fourth array holds weights.
__global__ void k_gather_sources(int n, int npad, const int* __restrict__ order,
                                 const double* __restrict__ x, const double* __restrict__ y,
                                 const double* __restrict__ z, const double* __restrict__ m,
                                 double G, const int* __restrict__ uniform,
                                 double* __restrict__ sx, double* __restrict__ sy,
                                 double* __restrict__ sz, double* __restrict__ sgm) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npad) return;
    const int i = order[p < n ? p : n - 1];
    sx[p] = x[i];
    sy[p] = y[i];
    sz[p] = z[i];
    const double w = (uniform && uniform[0]) ? 1.0 : G * m[i];
    sgm[p] = (p < n) ? w : 0.0;
}

// flag[0] must be 1 on entry; cleared when any mass differs from m[0].  common_gm[0] = G*m[0].
__global__ void k_mass_uniform(int n, const double* __restrict__ m, double G, int* __restrict__ flag,
                               double* __restrict__ common_gm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) common_gm[0] = G * m[0];
    if (i < n && m[i] != m[0]) flag[0] = 0;
}

// A coordinate one box length outside [0, L) is brought back in (x == L becomes 0), so that
// the cell a particle is binned in and the coordinate the pair loop uses always agree.
__device__ __forceinline__ double canonical(double x, double L) {
    x = (x >= L) ? x - L : x;
    x = (x < 0.0) ? x + L : x;
    return (x >= L) ? 0.0 : x;
}

__global__ void k_gather_targets(int n, int npad, const int* __restrict__ order,
                                 const double* __restrict__ x, const double* __restrict__ y,
                                 const double* __restrict__ z, double* __restrict__ tx,
                                 double* __restrict__ ty, double* __restrict__ tz) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npad) return;
    const int i = order[p < n ? p : n - 1];
    tx[p] = x[i];
    ty[p] = y[i];
    tz[p] = z[i];
}

// Conservative integer bounding box of each run of SUB sorted sources, in units of
// L / 2^QBITS: box = {xlo, xhi, ylo, yhi, zlo, zhi, 0, 0} with lo <= true min and
// hi > true max (one unit of slack each side absorbs the rounding of x * qscale).
__device__ __forceinline__ int quantize(double x, double qscale) {
    double q = floor(x * qscale);
    q = fmin(fmax(q, -1.0e9), 1.0e9);
    return static_cast<int>(q);
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}

__global__ void __launch_bounds__(WG) k_source_boxes(int npad, const double* __restrict__ sx,
                                                     const double* __restrict__ sy,
                                                     const double* __restrict__ sz,
                                                     double qscale, int* __restrict__ box) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;  // npad is a multiple of WG
    if (p >= npad) return;
    const int qx = quantize(sx[p], qscale), qy = quantize(sy[p], qscale),
              qz = quantize(sz[p], qscale);
    const int xlo = wave_min(qx) - 1, xhi = wave_max(qx) + 2;
    const int ylo = wave_min(qy) - 1, yhi = wave_max(qy) + 2;
    const int zlo = wave_min(qz) - 1, zhi = wave_max(qz) + 2;
    if ((threadIdx.x & 63) == 0) {
        int* b = box + (p >> 6) * 8;
        b[0] = xlo; b[1] = xhi; b[2] = ylo; b[3] = yhi; b[4] = zlo; b[5] = zhi;
        b[6] = 0; b[7] = 0;
    }
}

// ---------------------------------------------------------------------------------
// The pair kernel.
//
// One workgroup = 4 wavefronts.  Each lane keeps I targets in registers (position +
// three accumulators); a wavefront therefore owns 64*I consecutive sorted targets,
// a spatially compact group.  Sources stream through LDS in tiles of TJ 32-byte
// records {x, y, z, G*m}; every lane reads the same record (a broadcast ds_read_b128
// pair), so the inner loop is VALU work only.  The next tile's global loads are issued
// before the current tile is consumed (double-buffered LDS, one barrier per tile).
//
// Minimum image without per-pair work: for each (wavefront, 64-source sub-tile) the two
// integer bounding boxes decide, per axis, whether every pair of the two groups has the
// same image shift (0, +L or -L).  If so the shift is applied once to the wavefront's
// target coordinates and the pair loop does no wrap at all; otherwise the sub-tile is
// processed with the reference's per-pair compare/select (src/math_utils.h:11-16).
// Both give the reference's choice of image for every pair: the classification only
// accepts a uniform shift when all displacements are at least L/2^20 away from +-L/2.
//
//   MODE 0  direct summation, classified tiles
//   MODE 1  direct summation, per-pair wrap everywhere
//   MODE 2  P3M pair function over all pairs, per-pair wrap (small n / testing)
// ---------------------------------------------------------------------------------
// One sub-tile whose displacements along AXIS straddle +L/2 (sigma = +1: all x_j - x_i in
// [L/4, L)) or -L/2 (sigma = -1: all in (-L, -L/4]); the other two axes have a uniform image
// shift.  With d0 = x_j - x_i (the reference's first rounding, src/math_utils.h:11-16 is
// applied to this value), e = L/2 - sigma*d0 is exact (Sterbenz), and
//     w = copysign(L/2, e) - e
// equals sigma * wrap(d0) bit for bit: e >= +0 keeps d0 (including d0 == +-L/2, which the
// reference leaves unwrapped), e < 0 gives sigma*(d0 - sigma*L).  The force component along
// AXIS is accumulated in the sigma-flipped frame (accumulator multiplied by sigma before and
// after, both exact).
template <int I, int AXIS, bool SOFT0, bool WEIGHTED>
__device__ __forceinline__ void straddle_tile(const double4* __restrict__ src, const PairConsts& k,
                                              double sigma, const double (&shift)[3],
                                              const double (&xi)[I], const double (&yi)[I],
                                              const double (&zi)[I], double (&ax)[I],
                                              double (&ay)[I], double (&az)[I]) {
    double xs[I], ys[I], zs[I];
    const double nsig = -sigma;
    const double half = k.half_L;
#pragma unroll
    for (int u = 0; u < I; ++u) {
        xs[u] = (AXIS == 0) ? xi[u] : xi[u] + shift[0];
        ys[u] = (AXIS == 1) ? yi[u] : yi[u] + shift[1];
        zs[u] = (AXIS == 2) ? zi[u] : zi[u] + shift[2];
        if (AXIS == 0) ax[u] *= sigma;
        if (AXIS == 1) ay[u] *= sigma;
        if (AXIS == 2) az[u] *= sigma;
    }
#pragma unroll 2
    for (int j = 0; j < SUB; ++j) {
        const double4 pj = src[j];
#pragma unroll
        for (int u = 0; u < I; ++u) {
            double d[3] = {pj.x - xs[u], pj.y - ys[u], pj.z - zs[u]};
            const double e = __builtin_fma(nsig, d[AXIS], half);
            d[AXIS] = __builtin_copysign(half, e) - e;
            pair_direct<SOFT0, WEIGHTED>(d[0], d[1], d[2], pj.w, k, ax[u], ay[u], az[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < I; ++u) {
        if (AXIS == 0) ax[u] *= sigma;
        if (AXIS == 1) ay[u] *= sigma;
        if (AXIS == 2) az[u] *= sigma;
    }
}

struct PairArgs {
    const double* sx; const double* sy; const double* sz; const double* sgm;
    const int* sbox;      // [n_src_pad / SUB][8]
    int n_src_pad;        // multiple of TJ
    const double* tx; const double* ty; const double* tz;
    const int* tperm;     // sorted target -> index into ax/ay/az
    int n_tgt;
    int tiles_per_split;  // source tiles per blockIdx.y slice
    double* ax; double* ay; double* az;       // output (accumulated into) when gridDim.y == 1
    double* px; double* py; double* pz;       // partial sums [gridDim.y][n_tgt_pad] otherwise
    int n_tgt_pad;
    const int* uniform_mass;           // device flag: 1 when every source has the same mass; then
                                       // sgm holds weights (1 real, 0 padding) and the sums are
                                       // scaled by common_gm[0] at the end
    const double* common_gm;
    unsigned long long* tile_counts;   // optional: [0] (wavefront, sub-tile) visits, [1] of them
                                       // processed with the per-pair wrap
    PairConsts k;
};

template <int I, int MODE, bool SOFT0>
__global__ void __launch_bounds__(WG, 4) k_pairs(const PairArgs a) {  // 4 waves/SIMD: <= 128 VGPRs
    __shared__ __attribute__((aligned(32))) double4 tile[2][TJ];
    __shared__ int boxes[2][(TJ / SUB) * 8];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int group_base = (blockIdx.x * (WG / 64) + wave) * (64 * I);
    const PairConsts k = a.k;

    double xi[I], yi[I], zi[I], ax[I], ay[I], az[I];
    int ilo[3], ihi[3];
    {
        int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
#pragma unroll
        for (int t = 0; t < I; ++t) {
            int idx = group_base + t * 64 + lane;
            idx = idx < a.n_tgt ? idx : a.n_tgt - 1;
            idx = idx < 0 ? 0 : idx;
            xi[t] = a.tx[idx]; yi[t] = a.ty[idx]; zi[t] = a.tz[idx];
            ax[t] = 0.0; ay[t] = 0.0; az[t] = 0.0;
            const int q[3] = {quantize(xi[t], k.qscale), quantize(yi[t], k.qscale),
                              quantize(zi[t], k.qscale)};
#pragma unroll
            for (int c = 0; c < 3; ++c) { lo[c] = min(lo[c], q[c]); hi[c] = max(hi[c], q[c]); }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ilo[c] = __builtin_amdgcn_readfirstlane(wave_min(lo[c]) - 1);
            ihi[c] = __builtin_amdgcn_readfirstlane(wave_max(hi[c]) + 2);
        }
    }

    const int n_tiles = a.n_src_pad / TJ;
    const int tile_begin = blockIdx.y * a.tiles_per_split;
    int tile_end = tile_begin + a.tiles_per_split;
    tile_end = tile_end < n_tiles ? tile_end : n_tiles;

    unsigned n_visits = 0, n_slow = 0;
    const bool uniform = (MODE != 2) && __builtin_amdgcn_readfirstlane(a.uniform_mass[0]) != 0;

    // prologue: stage the first tile
    double4 stage;
    int stage_box = 0;
    if (tile_begin < tile_end) {
        const int p = tile_begin * TJ + threadIdx.x;
        stage = make_double4(a.sx[p], a.sy[p], a.sz[p], a.sgm[p]);
        if (threadIdx.x < (TJ / SUB) * 8) stage_box = a.sbox[tile_begin * (TJ / SUB) * 8 + threadIdx.x];
    }

    for (int t = tile_begin; t < tile_end; ++t) {
        const int buf = (t - tile_begin) & 1;
        const bool weighted = !uniform || t == n_tiles - 1;  // only the last tile can hold padding
        tile[buf][threadIdx.x] = stage;
        if (threadIdx.x < (TJ / SUB) * 8) boxes[buf][threadIdx.x] = stage_box;
        __syncthreads();
        if (t + 1 < tile_end) {  // prefetch the next tile into registers
            const int p = (t + 1) * TJ + threadIdx.x;
            stage = make_double4(a.sx[p], a.sy[p], a.sz[p], a.sgm[p]);
            if (threadIdx.x < (TJ / SUB) * 8) stage_box = a.sbox[(t + 1) * (TJ / SUB) * 8 + threadIdx.x];
        }

#pragma unroll 1
        for (int s = 0; s < TJ / SUB; ++s) {
            const double4* __restrict__ src = &tile[buf][s * SUB];
            // Per axis: 0 = every pair has the same image (shift known), 1 = displacements
            // straddle +L/2 or -L/2 (sigma says which), 2 = anything else.
            int n_straddle = 0, n_generic = 0, straddle_axis = 0;
            double shift[3] = {0.0, 0.0, 0.0};
            double sigma = 1.0;
            if constexpr (MODE == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int jlo = __builtin_amdgcn_readfirstlane(boxes[buf][s * 8 + 2 * c]);
                    const int jhi = __builtin_amdgcn_readfirstlane(boxes[buf][s * 8 + 2 * c + 1]);
                    const int dlo = jlo - ihi[c];  // < every x_j - x_i of the two groups
                    const int dhi = jhi - ilo[c];  // > every x_j - x_i
                    if (dhi <= QHALF - 1 && dlo >= -(QHALF - 1)) {
                        shift[c] = 0.0;            // no pair wraps
                    } else if (dlo >= QHALF + 1) {
                        shift[c] = k.L;            // every pair: d - L == x_j - (x_i + L)
                    } else if (dhi <= -(QHALF + 1)) {
                        shift[c] = -k.L;           // every pair: d + L == x_j - (x_i - L)
                    } else if (dlo >= QHALF / 2) {
                        ++n_straddle; straddle_axis = c; sigma = 1.0;    // all d in [L/4, L)
                    } else if (dhi <= -(QHALF / 2)) {
                        ++n_straddle; straddle_axis = c; sigma = -1.0;   // all d in (-L, -L/4]
                    } else {
                        ++n_generic;
                    }
                }
            } else {
                n_generic = 3;
            }
            ++n_visits;
            if (n_generic == 0 && n_straddle == 0) {
                double xs[I], ys[I], zs[I];
#pragma unroll
                for (int u = 0; u < I; ++u) {
                    xs[u] = xi[u] + shift[0]; ys[u] = yi[u] + shift[1]; zs[u] = zi[u] + shift[2];
                }
                if (weighted) {
#pragma unroll 2
                    for (int j = 0; j < SUB; ++j) {
                        const double4 pj = src[j];
#pragma unroll
                        for (int u = 0; u < I; ++u)
                            pair_direct<SOFT0, true>(pj.x - xs[u], pj.y - ys[u], pj.z - zs[u], pj.w,
                                                     k, ax[u], ay[u], az[u]);
                    }
                } else {
#pragma unroll 2
                    for (int j = 0; j < SUB; ++j) {
                        const double4 pj = src[j];
#pragma unroll
                        for (int u = 0; u < I; ++u)
                            pair_direct<SOFT0, false>(pj.x - xs[u], pj.y - ys[u], pj.z - zs[u], pj.w,
                                                      k, ax[u], ay[u], az[u]);
                    }
                }
            } else if (n_generic == 0 && n_straddle == 1) {
                ++n_slow;
                if (weighted) {
                    if (straddle_axis == 0)
                        straddle_tile<I, 0, SOFT0, true>(src, k, sigma, shift, xi, yi, zi, ax, ay, az);
                    else if (straddle_axis == 1)
                        straddle_tile<I, 1, SOFT0, true>(src, k, sigma, shift, xi, yi, zi, ax, ay, az);
                    else
                        straddle_tile<I, 2, SOFT0, true>(src, k, sigma, shift, xi, yi, zi, ax, ay, az);
                } else {
                    if (straddle_axis == 0)
                        straddle_tile<I, 0, SOFT0, false>(src, k, sigma, shift, xi, yi, zi, ax, ay, az);
                    else if (straddle_axis == 1)
                        straddle_tile<I, 1, SOFT0, false>(src, k, sigma, shift, xi, yi, zi, ax, ay, az);
                    else
                        straddle_tile<I, 2, SOFT0, false>(src, k, sigma, shift, xi, yi, zi, ax, ay, az);
                }
            } else {
                ++n_slow;
#pragma unroll 2
                for (int j = 0; j < SUB; ++j) {
                    const double4 pj = src[j];
#pragma unroll
                    for (int u = 0; u < I; ++u) {
                        const double dx = min_image(pj.x - xi[u], k.L, k.half_L);
                        const double dy = min_image(pj.y - yi[u], k.L, k.half_L);
                        const double dz = min_image(pj.z - zi[u], k.L, k.half_L);
                        if constexpr (MODE == 2)
                            pair_p3m(dx, dy, dz, pj.w, k, ax[u], ay[u], az[u]);
                        else
                            pair_direct<SOFT0>(dx, dy, dz, pj.w, k, ax[u], ay[u], az[u]);
                    }
                }
            }
        }
    }

    if (a.tile_counts && lane == 0) {
        atomicAdd(&a.tile_counts[0], static_cast<unsigned long long>(n_visits));
        atomicAdd(&a.tile_counts[1], static_cast<unsigned long long>(n_slow));
    }

    // epilogue: scatter to the caller's order, or store this slice's partial sums
    const double out_scale = uniform ? a.common_gm[0] : 1.0;
#pragma unroll
    for (int t = 0; t < I; ++t) {
        const int idx = group_base + t * 64 + lane;
        ax[t] *= out_scale; ay[t] *= out_scale; az[t] *= out_scale;
        if (idx < a.n_tgt) {
            if (gridDim.y == 1) {
                const int o = a.tperm[idx];
                a.ax[o] += ax[t]; a.ay[o] += ay[t]; a.az[o] += az[t];
            } else {
                const size_t o = static_cast<size_t>(blockIdx.y) * a.n_tgt_pad + idx;
                a.px[o] = ax[t]; a.py[o] = ay[t]; a.pz[o] = az[t];
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// P3M short range over a cell list.
//
// Search grid: `nside` cells per axis with R*cell >= cutoff radius, so all partners of a
// target lie in the (2R+1)^3 cells around its own cell (the reference's sweep:
// src/particles.cpp:187-197 / :287-297 with R = ceil(CUTOFF_RADIUS / CELL_SIZE), :134-137).
// The grid is stored with R ghost layers on every side: a particle within R cells of a face
// is entered a second (up to eighth) time in the ghost cell beyond the opposite face, with its
// coordinate already shifted by +-L.  The pair kernel therefore never wraps anything: the image
// implied by the ghost cell is the minimum image because nside >= 2R+3 keeps (R+1)*cell < L/2.
// Cell ids are z-fastest, id = (ix*np + iy)*np + iz, so the 2R+1 cells of one (dx,dy) row are
// one contiguous run of the sorted record array.  (The reference numbers its cells x-fastest,
// src/particles.cpp:94; z-fastest matches the order in which its initial conditions enumerate
// particles, src/ics.cpp:174-187, which keeps the binning passes close to sequential.)
//
// An entry of the sorted array is the particle's original index; its image code (2 bits per
// axis: 0 none, 1 = +L, 2 = -L) sits in a byte array beside it.  All entries of one cell carry
// the same code (it only depends on which side of the box the cell lies), so ordering a cell's
// entries leaves the codes in place.
// ---------------------------------------------------------------------------------

struct GhostGrid {
    double scale;   // nside / L
    double L;
    int nside;      // real cells per axis
    int R;          // ghost layers = search radius in cells
    int np;         // nside + 2R
    // Sources are binned only if their x cell lies in the circular range [keep_lo, keep_lo +
    // keep_len) of the nside cells: the x cells the targets occupy, widened by R on both sides.
    // keep_len = nside keeps everything (every particle is a target, or the targets are spread
    // over the whole box).  A rank of a multi-GPU run, whose targets are a slab, bins its slab
    // and a halo instead of the whole box.
    int keep_lo;
    int keep_len;
};

__device__ __forceinline__ bool x_cell_kept(const GhostGrid& g, int ix) {
    int d = ix - g.keep_lo;
    d += d < 0 ? g.nside : 0;
    return d < g.keep_len;
}

// occupied[ix] = 1 for the x cell of every target (array zeroed beforehand).  Plain stores of the
// same value: millions of targets share a few dozen cells, and atomics on them would serialise.
__global__ void k_target_x_cells(int n_tgt, const double* __restrict__ x, GhostGrid g, int* __restrict__ occupied) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tgt) return;
    occupied[bin_coord(canonical(x[i], g.L), g.scale, g.nside)] = 1;
}

// Calls f(padded_cell_id, code) for every image of a particle at canonical (x, y, z).
template <class F>
__device__ __forceinline__ void for_each_image(double x, double y, double z, const GhostGrid g, F f) {
    const int c[3] = {bin_coord(x, g.scale, g.nside), bin_coord(y, g.scale, g.nside),
                      bin_coord(z, g.scale, g.nside)};
    int idx[3][2], code[3][2], cnt[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        idx[a][0] = c[a] + g.R; code[a][0] = 0; cnt[a] = 1;
        if (c[a] < g.R) { idx[a][1] = c[a] + g.nside + g.R; code[a][1] = 1; cnt[a] = 2; }
        else if (c[a] >= g.nside - g.R) { idx[a][1] = c[a] - g.nside + g.R; code[a][1] = 2; cnt[a] = 2; }
    }
    for (int iz = 0; iz < cnt[2]; ++iz)
        for (int iy = 0; iy < cnt[1]; ++iy)
            for (int ix = 0; ix < cnt[0]; ++ix)
                f((idx[0][ix] * g.np + idx[1][iy]) * g.np + idx[2][iz],
                  static_cast<unsigned>(code[0][ix] | (code[1][iy] << 2) | (code[2][iz] << 4)));
}

// Counts every image's cell, the positions brought into [0, L) first (every pass after this one
// does the same on the fly: two compares per coordinate instead of three more arrays).  What the
// counter of its own cell held when a particle arrived is kept as the particle's rank there, so
// that k_ghost_scatter can place the real images without a second round of atomics.
__global__ void k_ghost_count(int n, const double* __restrict__ x, const double* __restrict__ y,
                              const double* __restrict__ z, GhostGrid g, int* __restrict__ count,
                              int* __restrict__ rank) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double px = canonical(x[i], g.L), py = canonical(y[i], g.L), pz = canonical(z[i], g.L);
    if (!x_cell_kept(g, bin_coord(px, g.scale, g.nside))) return;
    for_each_image(px, py, pz, g, [&](int cell, unsigned code) {
        if (code == 0u) rank[i] = atomicAdd(&count[cell], 1);
        else atomicAdd(&count[cell], 1);
    });
}

// out[0] += sum over cells of count^2 (out[0] zeroed beforehand); out[1] = number of entries.
// sum(count^2) / entries is the mean occupancy of the cell an entry sits in: about 1.6 for one
// particle per cell spread evenly, tens to hundreds once the particles have clustered.
__global__ void __launch_bounds__(WG) k_occupancy(int ncell, const int* __restrict__ count,
                                                  const int* __restrict__ start,
                                                  unsigned long long* __restrict__ out) {
    __shared__ unsigned long long lds[WG / 64];
    unsigned long long s = 0;
    for (int c = blockIdx.x * WG + threadIdx.x; c < ncell; c += gridDim.x * WG) {
        const unsigned long long k = static_cast<unsigned long long>(count[c]);
        s += k * k;
    }
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = lds[0] + lds[1] + lds[2] + lds[3];
        if (t) atomicAdd(&out[0], t);
        if (blockIdx.x == 0) out[1] = static_cast<unsigned long long>(start[ncell]);
    }
}

// Real images go to the rank k_ghost_count noted.  Ghost cells (which hold ghost images only) are
// filled by counting their counters in `fill` down to zero, a cell's slots being handed out from
// its last to its first.  The within-cell order produced here is arbitrary and is made
// deterministic by k_cell_canon / k_cell_canon_big afterwards.
__global__ void k_ghost_scatter(int n, const double* __restrict__ x, const double* __restrict__ y,
                                const double* __restrict__ z, GhostGrid g,
                                const int* __restrict__ start, int* __restrict__ fill,
                                const int* __restrict__ rank,
                                int* __restrict__ entries, uint8_t* __restrict__ codes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double px = canonical(x[i], g.L), py = canonical(y[i], g.L), pz = canonical(z[i], g.L);
    if (!x_cell_kept(g, bin_coord(px, g.scale, g.nside))) return;   // the same test as k_ghost_count's
    for_each_image(px, py, pz, g, [&](int cell, unsigned code) {
        const int slot = start[cell] + (code == 0u ? rank[i] : atomicSub(&fill[cell], 1) - 1);
        entries[slot] = i;
        codes[slot] = static_cast<uint8_t>(code);
    });
}

// The pair-once kernel's filter reads every candidate's position as three 16-bit integers: x and y
// in one word, z in a half word.  The kernel fetches each lane's own candidates - a gather, which the
// texture-address path issues at a fixed rate per instruction however few bytes come back - and
// fp32 coordinates took six 16-byte fetches per trip of eight candidates where these take three.
// q = round((coordinate - lo) * scale) with [lo, lo + 65535 / scale] covering the ghost layers;
// two positions a distance r apart differ by at most r * scale + sqrt(3) in these units (each
// component is off by at most one), which is what the integer bound allows for.
struct SymQuant {
    double lo, scale;
};
__device__ __forceinline__ unsigned sym_quantise(double c, const SymQuant& q) {
    const double v = (c - q.lo) * q.scale + 0.5;
    return static_cast<unsigned>(fmin(fmax(v, 0.0), 65535.0));
}

__device__ __forceinline__ double image_shift(unsigned code2, double L) {
    return code2 == 1u ? L : (code2 == 2u ? -L : 0.0);
}

// QUANT = false: the fp32 copies of the positions, one array per axis, for the cutoff pre-filter of
// k_p3m_cells, k_p3m_pool and k_p3m_rows_pool.  QUANT = true: 16-bit integer positions for k_p3m_sym's,
// x | y << 16 in `qxy`, z in `qz` (the float arrays are not written).
template <bool QUANT>
__global__ void k_ghost_gather(int n_entries, const int* __restrict__ entries,
                               const uint8_t* __restrict__ codes, const double* __restrict__ x, const double* __restrict__ y,
                               const double* __restrict__ z, const double* __restrict__ m,
                               double G, double L, double4* __restrict__ rec,
                               float* __restrict__ fx, float* __restrict__ fy, float* __restrict__ fz,
                               SymQuant q, uint32_t* __restrict__ qxy, uint16_t* __restrict__ qz) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_entries) return;
    const int i = entries[p];
    const unsigned code = codes[p];
    const double4 r = make_double4(canonical(x[i], L) + image_shift(code & 3u, L),
                                   canonical(y[i], L) + image_shift((code >> 2) & 3u, L),
                                   canonical(z[i], L) + image_shift((code >> 4) & 3u, L), G * m[i]);
    rec[p] = r;
    if constexpr (QUANT) {
        qxy[p] = sym_quantise(r.x, q) | (sym_quantise(r.y, q) << 16);
        qz[p] = static_cast<uint16_t>(sym_quantise(r.z, q));
    } else {
        fx[p] = static_cast<float>(r.x);
        fy[p] = static_cast<float>(r.y);
        fz[p] = static_cast<float>(r.z);
    }
}

__global__ void k_target_flags(int n_entries, const int* __restrict__ entries,
                               const uint8_t* __restrict__ codes, int tgt_begin, int n_tgt,
                               int* __restrict__ flag) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > n_entries) return;
    int f = 0;
    if (p < n_entries) {
        const int i = entries[p];
        f = (codes[p] == 0 && i >= tgt_begin && i - tgt_begin < n_tgt) ? 1 : 0;
    }
    flag[p] = f;  // entry n_entries is the trailing zero the scan needs
}

__global__ void k_target_scatter(int n_entries, const int* __restrict__ flag,
                                 const int* __restrict__ slot, int* __restrict__ tlist) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_entries && flag[p]) tlist[slot[p]] = p;
}

// When every particle is a target the list is the real entries in sorted order, and those are
// one contiguous piece per z column of the padded grid (cells R .. np-R-1 of a column that is
// not itself a ghost column).  count[col] = real entries of column col; count[np^2] = 0.
__global__ void k_column_counts(GhostGrid g, const int* __restrict__ cell_start, int* __restrict__ count) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int ncol = g.np * g.np;
    if (col > ncol) return;
    int c = 0;
    if (col < ncol) {
        const int X = col / g.np, Y = col - X * g.np;
        if (X >= g.R && X < g.np - g.R && Y >= g.R && Y < g.np - g.R)
            c = cell_start[col * g.np + g.np - g.R] - cell_start[col * g.np + g.R];
    }
    count[col] = c;
}

// The two steps above (counts, exclusive scan into col_start[0 .. np^2]) in one launch of one
// workgroup of 1024 threads, for grids of up to COLUMN_STARTS_SMALL_MAX columns, where four launches
// cost more than the work.  A thread owns COLS consecutive columns and requests all their bounds
// before it uses any, so the kernel waits for memory once (walking the columns 1024 at a time, one
// dependent pair of loads per round, took 46 us at 11 000 columns and three times that beside
// other kernels).
constexpr int COLUMN_STARTS_SMALL_COLS = 17;
constexpr int COLUMN_STARTS_SMALL_MAX = 1024 * COLUMN_STARTS_SMALL_COLS - 1;   // entry ncol is the total
__global__ void __launch_bounds__(1024) k_column_starts_small(GhostGrid g, const int* __restrict__ cell_start,
                                                              int* __restrict__ col_start) {
    constexpr int COLS = COLUMN_STARTS_SMALL_COLS;
    __shared__ int wave_total[16];
    const int ncol = g.np * g.np;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int first = threadIdx.x * COLS;
    int lo[COLS], hi[COLS];
#pragma unroll
    for (int k = 0; k < COLS; ++k) {
        const int col = first + k;
        const int X = col / g.np, Y = col - X * g.np;
        const bool real = col < ncol && X >= g.R && X < g.np - g.R && Y >= g.R && Y < g.np - g.R;
        lo[k] = real ? cell_start[col * g.np + g.R] : 0;
        hi[k] = real ? cell_start[col * g.np + g.np - g.R] : 0;
    }
    int mine = 0;
#pragma unroll
    for (int k = 0; k < COLS; ++k) mine += hi[k] - lo[k];
    int inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wave_total[wave] = inc;
    __syncthreads();
    int at = inc - mine;
    for (int w = 0; w < wave; ++w) at += wave_total[w];
#pragma unroll
    for (int k = 0; k < COLS; ++k) {
        const int col = first + k;
        if (col <= ncol) col_start[col] = at;
        at += hi[k] - lo[k];
    }
}

// col_start = exclusive scan of the counts above: the list positions of a column's real entries
__global__ void __launch_bounds__(WG) k_column_targets(GhostGrid g, const int* __restrict__ cell_start,
                                                       const int* __restrict__ col_start,
                                                       int* __restrict__ tlist) {
    const int ncol = g.np * g.np;
    for (int col = blockIdx.x; col < ncol; col += gridDim.x) {
        const int at = col_start[col], cnt = col_start[col + 1] - at;
        if (cnt == 0) continue;
        const int first = cell_start[col * g.np + g.R];
        for (int k = threadIdx.x; k < cnt; k += WG) tlist[at + k] = first + k;
    }
}

// The accepted-pair arithmetic of src/particles.cpp:330-359 for one (target, source) pair with
// r^2 <= cutoff_sq already established.
template <bool SOFT0>
__device__ __forceinline__ void p3m_accepted(double dx, double dy, double dz, double r2, double gm,
                                             const PairConsts& k, double& ax, double& ay, double& az) {
    auto cube = [&](double q) {
        const double y = __builtin_amdgcn_rsq(q);
        const double y2 = y * y;
        const double h = __builtin_fma(-q, y2, 1.0);
        const double y3 = y * y2;
        return __builtin_fma(y3, h * __builtin_fma(k.c1875, h, k.c15), y3);
    };
    const double q1 = r2 + k.eps2;
    double f1 = cube(q1);
    if constexpr (SOFT0) f1 = (q1 > 0.0) ? f1 : 0.0;  // coincident particles, zero softening
    double f = f1 - cube(r2 + k.eps_pm2);
    if (r2 > k.r_switch_sq) {
        const double dist = r2 * inv_sqrt_refined(r2);
        const double t = (dist - k.r_switch) * k.inv_width;
        f *= __builtin_fma(t * t, __builtin_fma(2.0, t, -3.0), 1.0);
    }
    const double s = gm * f;
    ax = __builtin_fma(s, dx, ax);
    ay = __builtin_fma(s, dy, ay);
    az = __builtin_fma(s, dz, az);
}

struct P3MArgs {
    const double4* rec;      // sorted records incl. ghosts: {x, y, z, G*m}
    const float* fx;         // the same positions rounded to fp32 (pre-filter), one array per
    const float* fy;         // axis, each n_entries + P3M_TRIP long
    const float* fz;
    float prefilter_sq;      // fp32 r^2 bound that no pair with r^2 <= cutoff_sq can exceed
    const uint32_t* qxy;     // k_p3m_sym instead of fx, fy, fz: positions as 16-bit integers (SymQuant), x | y << 16
    const uint16_t* qz;      // and z, each n_entries + 8 long
    int qbound;              // integer r^2 bound in those units that no pair inside the cutoff can exceed
    const int* entries;      // original index of every sorted record
    const int* cell_start;   // [np^3 + 1]
    const int* tlist;        // sorted positions of the targets
    int n_entries;
    int n_tgt;
    int tgt_begin;
    GhostGrid g;
    double* ax; double* ay; double* az;
    unsigned long long* accepted;   // optional counter of pairs with r^2 <= cutoff_sq
    PairConsts k;
};

// The accepted-pair counter is P3M_COUNT_STRIPES partial counters 128 bytes apart, one per
// workgroup number modulo the stripes: a few hundred thousand atomics on ONE address take
// longer than the pair kernel itself (about 11 ns each, one after the other, in the L2).
constexpr int P3M_COUNT_STRIPES = 64;
constexpr int P3M_COUNT_STRIDE = 16;   // in 8-byte words
__device__ __forceinline__ unsigned long long* accepted_stripe(unsigned long long* base) {
    return base + (blockIdx.x % P3M_COUNT_STRIPES) * P3M_COUNT_STRIDE;
}

constexpr int P3M_QCAP = 16;    // accepted-pair ring per lane (power of 2; 16 KiB LDS per workgroup)
constexpr int P3M_DRAIN_UNROLL = 2;   // record fetches in flight per lane while draining (1: 62 instead of 72
                                      // registers and an eighth wavefront per SIMD, yet 5 % slower; 4: 3 % slower)
constexpr int P3M_DRAIN = 8;   // entries every lane evaluates when some lane's ring is full
#ifndef NBPP_CELLS_WG
#define NBPP_CELLS_WG 256
#endif
constexpr int P3M_CELLS_WG = NBPP_CELLS_WG;   // threads per workgroup of k_p3m_cells (no workgroup barrier in it)
constexpr int P3M_XCD_RUN = 16;   // consecutive pieces of the target list one XCD takes (see k_p3m_cells)
constexpr int P3M_TRIP = 4;    // candidates a lane tests per trip of the filter loop (2, 4 or 8)

// One lane per target.  Two phases per lane, interleaved wave-wide:
//   filter: walk the (2R+1)^2 runs, P3M_TRIP candidates per trip, and push the sorted position
//           of every candidate that may lie inside the cutoff onto the lane's ring in LDS (~30 %
//           pass).  The test runs in single precision against a bound that no pair inside the
//           cutoff can exceed (P3MArgs::prefilter_sq).  The fp32 coordinates sit in one array per
//           axis, so a lane's P3M_TRIP consecutive candidates arrive as one vector per axis and
//           the arithmetic is packed (v_pk_*_f32, two candidates per instruction) without any
//           register shuffling; the loop is bound by instruction issue.
//   drain:  when any lane's ring is full every lane evaluates its P3M_DRAIN oldest entries from
//           the fp64 records: exact r^2, the reference's cutoff test (src/particles.cpp:326),
//           then the full taper / double-softening arithmetic.  What a lane does not evaluate
//           now stays queued; the rest is done at the end.
// Without the queue every candidate iteration in which ANY lane passes the cutoff pays for the
// full pair function on all 64 lanes.
//
// WIDE = false addresses records with 32-bit byte offsets (n_entries < 2^27).
template <bool WIDE>
__device__ __forceinline__ double4 load_record(const double4* rec, int idx) {
    if constexpr (WIDE) return rec[idx];
    return *reinterpret_cast<const double4*>(reinterpret_cast<const char*>(rec) +
                                             (static_cast<uint32_t>(idx) << 5));
}
__device__ __forceinline__ int load_word(const int* base, int idx) {   // idx < 2^30
    return *reinterpret_cast<const int*>(reinterpret_cast<const char*>(base) +
                                         (static_cast<uint32_t>(idx) << 2));
}
// Row bounds through a range-checked buffer load: a lane without a row passes an offset past
// the end and gets 0 in the destination register itself, with no select after the load, so the
// value is not waited for before it is used a row later.  CHECKED = false (the ncell + 1 words
// take 4 GiB or more): plain loads.
template <bool CHECKED>
struct StartTable {
    __amdgpu_buffer_rsrc_t rsrc;
    const int* base;
};
template <bool CHECKED>
__device__ __forceinline__ StartTable<CHECKED> make_start_table(const int* cell_start, const GhostGrid& g) {
    const unsigned long long bytes = (1ull * g.np * g.np * g.np + 1ull) << 2;
    return StartTable<CHECKED>{__builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(cell_start), 0,
                                                                 CHECKED ? static_cast<uint32_t>(bytes) : 0u, 0x00020000),
                               cell_start};
}
template <bool CHECKED>
__device__ __forceinline__ int load_start(const StartTable<CHECKED>& t, int idx, bool has) {
    if constexpr (CHECKED)
        return static_cast<int>(__builtin_amdgcn_raw_buffer_load_b32(t.rsrc, has ? static_cast<uint32_t>(idx) << 2 : 0xfffffffcu, 0, 0));
    else
        return has ? load_word(t.base, idx) : 0;
}
constexpr unsigned long long START_TABLE_CHECKED_MAX_CELLS = (0xfffffff0ull >> 2) - 1ull;   // cells whose table fits

static_assert((P3M_QCAP & (P3M_QCAP - 1)) == 0 && P3M_DRAIN % P3M_DRAIN_UNROLL == 0 && P3M_DRAIN % 4 == 0 &&
              P3M_QCAP % P3M_DRAIN_UNROLL == 0 && P3M_DRAIN >= P3M_TRIP && (P3M_TRIP == 2 || P3M_TRIP == 4 || P3M_TRIP == 8),
              "ring sizes");
__device__ __forceinline__ int ring_slot(int i) { return i & (P3M_QCAP - 1); }

// Loads that fetch nothing and return zeros for lanes whose byte offset is past the end of the
// buffer (hardware range check): masking a lane costs one select, no branch.
typedef float p3m_vf __attribute__((ext_vector_type(P3M_TRIP)));
template <int T> struct p3m_fvec_load;
template <> struct p3m_fvec_load<2> {
    typedef float vf __attribute__((ext_vector_type(2)));
    typedef unsigned vu __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ vf get(__amdgpu_buffer_rsrc_t rsrc, uint32_t off) {
        return __builtin_bit_cast(vf, (vu)__builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0));
    }
};
template <> struct p3m_fvec_load<8> {
    typedef float vf __attribute__((ext_vector_type(8)));
    typedef float hf __attribute__((ext_vector_type(4)));
    typedef unsigned vu __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ vf get(__amdgpu_buffer_rsrc_t rsrc, uint32_t off) {
        const hf lo = __builtin_bit_cast(hf, (vu)__builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
        const hf hi = __builtin_bit_cast(hf, (vu)__builtin_amdgcn_raw_buffer_load_b128(rsrc, off == 0xfffffff0u ? off : off + 16u, 0, 0));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
};
template <> struct p3m_fvec_load<4> {
    typedef float vf __attribute__((ext_vector_type(4)));
    typedef unsigned vu __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ vf get(__amdgpu_buffer_rsrc_t rsrc, uint32_t off) {
        return __builtin_bit_cast(vf, (vu)__builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    }
};
__device__ __forceinline__ p3m_vf load_fvec_checked(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_offset) {
    return p3m_fvec_load<P3M_TRIP>::get(rsrc, byte_offset);
}
// the same vector through a plain pointer (4-byte aligned)
__device__ __forceinline__ p3m_vf load_fvec(const float* base, int idx) {
    p3m_vf v;
#pragma unroll
    for (int u = 0; u < P3M_TRIP; ++u) v[u] = base[static_cast<size_t>(idx) + u];
    return v;
}

__device__ __forceinline__ double4 load_record_checked(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_offset) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d lo = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_offset, 0, 0));
    const v2d hi = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_offset + 16u, 0, 0));
    return make_double4(lo.x, lo.y, hi.x, hi.y);
}

// Of a row's 2R + 1 cells a target only has to walk those its cutoff sphere meets.  With
// (tx, ty, tz) the target's place inside its own cell (in cells, [0, 1]) and dm the distance
// from it to the nearest face of the row's column (ox, oy cells away), the sphere spans
// tz +- sqrt(rc^2 - dm^2) there, nothing when that is negative: 44 % fewer candidates at two
// cells per cutoff radius.  Single precision with a margin of 1e-4 cells (positions and cell
// numbers come from the same double-precision product, so they agree to far better than that).
// The lanes of a wavefront still walk a row for as long as the longest of their windows, so the
// time saved is a few per cent (k_p3m_pool, k_p3m_sym); k_p3m_cells, whose extra registers cost it
// a wavefront per SIMD, walks whole rows.
struct RowTrim {
    float tx, ty, tz;
    float rc2;      // (cutoff radius in cells)^2, rounded up
};
__device__ __forceinline__ RowTrim make_row_trim(const double4& me, int pcx, int pcy, int pcz, const GhostGrid& g,
                                                 const PairConsts& k) {
    const auto inside = [&](double c, int cell) {
        return fminf(fmaxf(static_cast<float>(c * g.scale - static_cast<double>(cell - g.R)), 0.0f), 1.0f);
    };
    const float rc = static_cast<float>(sqrt(k.cutoff_sq) * g.scale);
    return RowTrim{inside(me.x, pcx), inside(me.y, pcy), inside(me.z, pcz), rc * rc * (1.0f + 1e-5f)};
}
// cells [zlo, zhi] (relative to the target's) of the row (ox, oy) the sphere meets; false: none
__device__ __forceinline__ bool row_window(const RowTrim& w, int ox, int oy, int R, int& zlo, int& zhi) {
    const float dxm = ox > 0 ? static_cast<float>(ox) - w.tx : (ox < 0 ? w.tx - static_cast<float>(ox + 1) : 0.0f);
    const float dym = oy > 0 ? static_cast<float>(oy) - w.ty : (oy < 0 ? w.ty - static_cast<float>(oy + 1) : 0.0f);
    const float h2 = w.rc2 - dxm * dxm - dym * dym;
    const float h = __builtin_amdgcn_sqrtf(fmaxf(h2, 0.0f)) + 1e-4f;   // v_sqrt_f32 as it is: 1 ulp, far inside the margin
    zlo = max(-R, static_cast<int>(floorf(w.tz - h)));
    zhi = min(R, static_cast<int>(floorf(w.tz + h)));
    return h2 >= 0.0f;
}

// S (1, 2, 4 or 8) lanes share one target, lane `sub` taking rows sub, sub + S, ...; their sums
// are added in a fixed order at the end.  One lane per target cannot fill the GPU below a few
// hundred thousand targets (the reference's shipped run has 32768).
template <bool SOFT0, bool WIDE, int S>
__global__ void __launch_bounds__(P3M_CELLS_WG) k_p3m_cells(const P3MArgs a) {
    __shared__ int queue[P3M_QCAP][P3M_CELLS_WG];
    // Workgroups are handed to the 8 XCDs round-robin and every XCD has its own L2.  Neighbouring
    // pieces of the (cell-ordered) target list read nearly the same runs, so within every group of
    // 8 * P3M_XCD_RUN workgroups XCD k takes P3M_XCD_RUN consecutive pieces.  (Giving each XCD one
    // contiguous eighth of the whole list halves the HBM traffic again but puts a whole dense
    // region on one XCD: 25 % slower on a clustered set.)
    const int b = static_cast<int>(blockIdx.x);
    constexpr int RUN = P3M_XCD_RUN * (WG / P3M_CELLS_WG);
    const int group = b / (8 * RUN);
    const int in_group = b - group * (8 * RUN);
    const bool whole = (group + 1) * (8 * RUN) <= static_cast<int>(gridDim.x);
    const int piece = whole ? group * (8 * RUN) + (in_group % 8) * RUN + in_group / 8 : b;
    const int gt = piece * P3M_CELLS_WG + threadIdx.x;
    const int t = gt / S, sub = gt % S;
    const bool live = t < a.n_tgt;
    const int p = a.tlist[live ? t : a.n_tgt - 1];
    const double4 me = a.rec[p];
    const p3m_vf nmx = -a.fx[p], nmy = -a.fy[p], nmz = -a.fz[p];   // splats
    const GhostGrid g = a.g;
    const PairConsts k = a.k;
    const float bound = a.prefilter_sq;
    const int pcx = bin_coord(me.x, g.scale, g.nside) + g.R;
    const int pcy = bin_coord(me.y, g.scale, g.nside) + g.R;
    const int pcz = bin_coord(me.z, g.scale, g.nside) + g.R;
    double ax = 0.0, ay = 0.0, az = 0.0;
    int rp = 0, wp = 0;   // ring read / write positions, unwrapped: wp - rp entries are queued
    int lim = P3M_QCAP - P3M_TRIP;   // the filter stops for a drain once some lane's wp exceeds it
    unsigned long long hits = 0;
    // the arrays are P3M_TRIP values longer than n_entries, so a vector that starts at a valid
    // entry lies inside the range check.  WIDE: byte offsets may not fit 32 bits; plain pointer
    // loads then, exhausted lanes pointed at their own record.
    const uint32_t f_bytes = WIDE ? 0u : static_cast<uint32_t>(a.n_entries + P3M_TRIP) << 2;
    const __amdgpu_buffer_rsrc_t fx_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fx), 0, f_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t fy_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fy), 0, f_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t fz_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fz), 0, f_bytes, 0x00020000);

    // evaluates up to `quota` (a multiple of DU) of the oldest queued candidates of every lane,
    // DU at a time so that DU record fetches are in flight together: the loop is a chain of
    // LDS read -> record fetch -> arithmetic, and the fetch latency is most of its cost
    constexpr int DU = P3M_DRAIN_UNROLL;
    const uint32_t rec_bytes = WIDE ? 0xffffffe0u : static_cast<uint32_t>(a.n_entries) << 5;
    const __amdgpu_buffer_rsrc_t rec_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double4*>(a.rec), 0, rec_bytes, 0x00020000);
    auto drain = [&](int quota) {
        const int cnt = wp - rp;
#pragma unroll 1
        for (int q = 0; q < quota && __any(q < cnt); q += DU) {
            double4 s[DU];
            bool valid[DU];
#pragma unroll
            for (int u = 0; u < DU; ++u) {
                const int idx = queue[q + u < cnt ? ring_slot(rp + q + u) : 0][threadIdx.x];
                valid[u] = q + u < cnt && idx != p;   // the target itself passes the pre-filter
                if constexpr (WIDE) s[u] = a.rec[valid[u] ? idx : p];
                else s[u] = load_record_checked(rec_rsrc, valid[u] ? static_cast<uint32_t>(idx) << 5 : 0xffffffe0u);
            }
#pragma unroll
            for (int u = 0; u < DU; ++u) {
                const double dx = s[u].x - me.x, dy = s[u].y - me.y, dz = s[u].z - me.z;
                const double r2 = __builtin_fma(dx, dx, __builtin_fma(dy, dy, dz * dz));
                const bool accept = live && valid[u] && !(r2 > k.cutoff_sq);   // (padding lanes walk the last target's rows)
                hits += __popcll(__builtin_amdgcn_ballot_w64(accept));   // the wavefront's count, kept on the scalar side
                if (accept) p3m_accepted<SOFT0>(dx, dy, dz, r2, s[u].w, k, ax, ay, az);
            }
        }
        rp += min(quota, cnt);
        lim = rp + (P3M_QCAP - P3M_TRIP);
    };

    // Rows (ox, oy) in [-R, R]^2, oy fastest; the wavefront walks them together (S = 1) or S at a
    // time, one per lane of a target (S > 1).  `base` is the padded id of the middle cell of the
    // row whose bounds are in flight into (nj, nje): they are requested when the row before is
    // entered.
    const int span = 2 * g.R + 1;
    const int n_rows = span * span;
    const int n_rounds = (n_rows + S - 1) / S;
    const int row_jump = g.np * (g.np - span + 1);   // last row of one ox to the first of the next
    int base = ((pcx - g.R) * g.np + (pcy - g.R)) * g.np + pcz;   // row 0
    int round = 0, oy = 0;   // rounds entered so far; wave-uniform
    int j = 0, je = 0;
    int nj = 0, nje = 0;
    // bounds of this lane's row of round `r` (S > 1): row index r * S + sub, none past the end
    auto request_bounds = [&](int r) {
        const int my_row = r * S + sub;
        const int ox = my_row / span, oyy = my_row - ox * span;
        const int b = ((pcx - g.R + ox) * g.np + (pcy - g.R + oyy)) * g.np + pcz;
        const bool has = my_row < n_rows;
        nj = has ? load_word(a.cell_start, b - g.R) : 0;
        nje = has ? load_word(a.cell_start, b + g.R + 1) : 0;
    };
    if constexpr (S == 1) {
        nj = load_word(a.cell_start, base - g.R);
        nje = load_word(a.cell_start, base + g.R + 1);
    } else {
        request_bounds(0);
    }
    bool done = false;
#pragma unroll 1
    while (!done) {
        // ---- filter until some lane's ring is full or the rows are used up.  This loop does
        // not touch the sums, so they stay in their registers across it.
#pragma unroll 1
        for (;;) {
            if (!__any(j < je)) {   // every lane has finished its run: enter the next round of rows
                if (round == n_rounds) { done = true; break; }
                j = nj; je = nje;
                ++round;
                if constexpr (S == 1) {
                    const bool wrap = oy + 1 == span;
                    base += (round < n_rows) ? (wrap ? row_jump : g.np) : 0;   // after the last row: itself again
                    oy = wrap ? 0 : oy + 1;
                    nj = load_word(a.cell_start, base - g.R);
                    nje = load_word(a.cell_start, base + g.R + 1);
                } else {
                    request_bounds(round);
                }
            }
            // P3M_TRIP candidates per lane per trip; an exhausted lane fetches nothing
            p3m_vf cx, cy, cz;
            if constexpr (WIDE) {
                const int b = j < je ? j : p;
                cx = load_fvec(a.fx, b); cy = load_fvec(a.fy, b); cz = load_fvec(a.fz, b);
            } else {
                const uint32_t off = j < je ? static_cast<uint32_t>(j) << 2 : 0xfffffff0u;
                cx = load_fvec_checked(fx_rsrc, off);
                cy = load_fvec_checked(fy_rsrc, off);
                cz = load_fvec_checked(fz_rsrc, off);
            }
            const p3m_vf dx = cx + nmx, dy = cy + nmy, dz = cz + nmz;
            const p3m_vf r2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, dz * dz));
            // the slot after the newest entry is free (at most QCAP - P3M_TRIP queued on entry):
            // the stores need no branch
            const int left = je - j;          // candidates of this trip that belong to the run
#pragma unroll
            for (int u = 0; u < P3M_TRIP; ++u) {
                queue[ring_slot(wp)][threadIdx.x] = j + u;
                wp += ((u < left) & !(r2[u] > bound)) ? 1 : 0;
            }
            j += P3M_TRIP;
            if (__any(wp > lim)) break;
        }
        drain(done ? P3M_QCAP : P3M_DRAIN);
    }

    // the S lanes of a target sit next to each other: fixed-order tree sum
#pragma unroll
    for (int d = 1; d < S; d <<= 1) {
        ax += __shfl_xor(ax, d, 64);
        ay += __shfl_xor(ay, d, 64);
        az += __shfl_xor(az, d, 64);
    }
    if (live && sub == 0) {
        const int o = a.entries[p] - a.tgt_begin;
        a.ax[o] += ax; a.ay[o] += ay; a.az[o] += az;
    }
    if (a.accepted) {
        const unsigned long long h = hits;
        if ((threadIdx.x & 63) == 0 && h) atomicAdd(accepted_stripe(a.accepted), h);
    }
}

// ---------------------------------------------------------------------------------
// The same search with the survivors of the filter POOLED per wavefront.
//
// In k_p3m_cells every lane drains its own ring, so a wavefront walks the drain loop as often as
// its fullest lane needs: 85 iterations for a mean of 62-71 accepted pairs per target on a
// near-uniform set, twice the balanced count once the particles have clustered.  Here the
// filter is the same (one lane per target, P3M_TRIP candidates per trip), but every survivor goes
// as a (target lane, source) pair into one ring per wavefront: for each of the P3M_TRIP candidate
// slots the survivors of the 64 lanes are compacted with a ballot and a prefix count, so a slot's
// batch holds every target at most once.  Whenever the ring holds 2 * 64 pairs the wavefront
// evaluates them 64 at a time, two in flight per lane, one pair per lane whatever target it
// belongs to: every lane of every drain step does the full arithmetic.  The lane reads its
// pair's target position from a table in LDS and adds the pair's contribution to that target's
// sums in LDS (ds_add_f64; a step's 64 pairs come from at most three batches, so few lanes
// collide).  The order in which colliding lanes add is the hardware's: sums are not bitwise
// reproducible from run to run (NBPP_OPT_DETERMINISTIC keeps k_p3m_cells).
// ---------------------------------------------------------------------------------
constexpr int P3M_POOL_CAP = 512;   // pairs per wavefront ring (power of 2): < 128 left by a drain + 4 * 64 from half a trip
constexpr int P3M_POOL_TRIP = 8;    // candidates a lane tests per trip: the runs of a clustered set are long
                                    // (16: 120 registers, 4 waves/SIMD, slower on every set: a = 0.51 1.66 against 1.54 ms)
// Threads per workgroup of k_p3m_pool: the template parameter PWG.  Its wavefronts work
// independently (a ring, a target table and sums per wavefront, no workgroup barrier), so any
// multiple of 64 will do.  Run by itself the kernel is fastest with one wavefront per workgroup,
// whose slot on the CU is free again as soon as that wavefront is done (64 / 128 / 256 threads: the
// 128^3 run at a = 0.51 1.29 / 1.29 / 1.40 ms, at a = 0.21 0.746 / 0.756 / 0.773, 400 blobs 11.9 /
// 12.2 / 12.4 ms, the 256^3 lattice 4.77 / 4.73 / 4.74).  Inside a whole step, where mesh and gas
// kernels run beside it on another stream, it is the other way round: single wavefront slots are
// refilled by this kernel before a four-wavefront workgroup of the other stream ever fits, the
// mesh chain starves and the cycle takes 1.69 ms with 64 or 128 threads against 1.57 with 256
// (128^3 particles + 128^3 gas mesh).  Hence NBPP_OPT_P3M_POOL_WORKGROUP.
constexpr int P3M_POOL_LANE_SHIFT = 26;   // ring entry = target lane << 26 | sorted position of the source (< 2^26)
typedef float p3m_pool_vf __attribute__((ext_vector_type(P3M_POOL_TRIP)));

template <bool SOFT0, bool WIDE, int PWG>
__global__ void __launch_bounds__(PWG) k_p3m_pool(const P3MArgs a) {
    // ring entries: target lane << 26 | sorted position of the source; {lane, position} when positions need more bits
    using Entry = std::conditional_t<WIDE, uint2, unsigned>;
    __shared__ Entry pool[PWG / 64][P3M_POOL_CAP];
    __shared__ double2 txy[PWG];                      // target x, y
    __shared__ double tz[PWG];                        // target z
    __shared__ double tsum[3][PWG];                   // the targets' sums
    const int b = static_cast<int>(blockIdx.x);      // XCD-aware order of the pieces: see k_p3m_cells
    constexpr int RUN = P3M_XCD_RUN * (WG / PWG);   // the same stretch of the target list per XCD
    const int group = b / (8 * RUN);
    const int in_group = b - group * (8 * RUN);
    const bool whole = (group + 1) * (8 * RUN) <= static_cast<int>(gridDim.x);
    const int piece = whole ? group * (8 * RUN) + (in_group % 8) * RUN + in_group / 8 : b;
    const int t = piece * PWG + threadIdx.x;
    const bool live = t < a.n_tgt;
    const int p = a.tlist[live ? t : a.n_tgt - 1];
    const double4 me = a.rec[p];
    const p3m_pool_vf nmx = -a.fx[p], nmy = -a.fy[p], nmz = -a.fz[p];
    const GhostGrid g = a.g;
    const PairConsts k = a.k;
    const float bound = a.prefilter_sq;
    const int pcx = bin_coord(me.x, g.scale, g.nside) + g.R;
    const int pcy = bin_coord(me.y, g.scale, g.nside) + g.R;
    const int pcz = bin_coord(me.z, g.scale, g.nside) + g.R;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wave_base = wave << 6;
    Entry* const ring = pool[wave];
    txy[threadIdx.x] = make_double2(me.x, me.y);
    tz[threadIdx.x] = me.z;
    tsum[0][threadIdx.x] = 0.0;
    tsum[1][threadIdx.x] = 0.0;
    tsum[2][threadIdx.x] = 0.0;
    __builtin_amdgcn_wave_barrier();   // only this wavefront reads them back, and LDS operations of a wavefront stay in order
    int rp = 0, wp = 0;                // ring read / write positions, unwrapped, the same in every lane
    unsigned long long hits = 0;                 // accepted pairs of the wavefront (the same in every lane)
    const uint32_t f_bytes = WIDE ? 0u : static_cast<uint32_t>(a.n_entries + P3M_POOL_TRIP) << 2;
    const __amdgpu_buffer_rsrc_t fx_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fx), 0, f_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t fy_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fy), 0, f_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t fz_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fz), 0, f_bytes, 0x00020000);
    const uint32_t rec_bytes = WIDE ? 0xffffffe0u : static_cast<uint32_t>(a.n_entries) << 5;
    const __amdgpu_buffer_rsrc_t rec_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double4*>(a.rec), 0, rec_bytes, 0x00020000);

    // one drain step: NB * 64 pairs from the front of the ring, NB per lane in flight together
    // (ring read -> table read and record fetch -> arithmetic -> three LDS adds).  A target's
    // own record passes the filter once: its displacement is exactly zero, so it adds exactly
    // zero to the sums (with zero softening p3m_accepted returns 0 for r = 0); it is taken out of
    // the pair count at the end.
    // FULL: the step takes NB * 64 entries, every lane has NB pairs.
    auto drain = [&](auto nb_tag, auto full_tag, int count) {
        constexpr int NB = decltype(nb_tag)::value;
        constexpr bool FULL = decltype(full_tag)::value;
        unsigned tl[NB], sj[NB];      // target lane, source
        bool valid[NB];
        double4 s[NB];
        double2 gxy[NB];
        double gz[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            valid[u] = FULL || u * 64 + lane < count;
            const Entry e = ring[(rp + u * 64 + lane) & (P3M_POOL_CAP - 1)];
            if constexpr (WIDE) {
                tl[u] = valid[u] ? e.x : static_cast<unsigned>(lane);
                sj[u] = valid[u] ? e.y : static_cast<unsigned>(p);
            } else {
                tl[u] = valid[u] ? e >> P3M_POOL_LANE_SHIFT : static_cast<unsigned>(lane);
                sj[u] = valid[u] ? e & ((1u << P3M_POOL_LANE_SHIFT) - 1u) : static_cast<unsigned>(p);
            }
            gxy[u] = txy[wave_base + static_cast<int>(tl[u])];
            gz[u] = tz[wave_base + static_cast<int>(tl[u])];
            if constexpr (WIDE) s[u] = a.rec[static_cast<int>(sj[u])];
            else s[u] = load_record_checked(rec_rsrc, valid[u] ? sj[u] << 5 : 0xffffffe0u);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const double dx = s[u].x - gxy[u].x, dy = s[u].y - gxy[u].y, dz = s[u].z - gz[u];
            const double r2 = __builtin_fma(dx, dx, __builtin_fma(dy, dy, dz * dz));
            const bool accept = valid[u] && !(r2 > k.cutoff_sq);   // the reference's cutoff test (src/particles.cpp:326)
            hits += __popcll(__builtin_amdgcn_ballot_w64(accept));   // counted on the scalar side
            if (accept) {
                double cx = 0.0, cy = 0.0, cz = 0.0;
                p3m_accepted<SOFT0>(dx, dy, dz, r2, s[u].w, k, cx, cy, cz);
                const int slot = wave_base + static_cast<int>(tl[u]);
                __hip_atomic_fetch_add(&tsum[0][slot], cx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&tsum[1][slot], cy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&tsum[2][slot], cz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        rp += FULL ? NB * 64 : min(count, NB * 64);
    };

    const int span = 2 * g.R + 1;
    const int n_rows = span * span;
    const int row_jump = g.np * (g.np - span + 1);
    int base = ((pcx - g.R) * g.np + (pcy - g.R)) * g.np + pcz;
    int round = 0, ox = 0, oy = 0;
    int j = 0, je = 0;
    int nj = 0, nje = 0;
    const RowTrim trim = make_row_trim(me, pcx, pcy, pcz, g, k);
    const StartTable<!WIDE> starts = make_start_table<!WIDE>(a.cell_start, g);
    // the part of row (ox, oy) this target can reach; padding lanes copy the last target: their runs are empty
    auto request_row = [&](bool has) {
        int zlo, zhi;
        has = row_window(trim, ox - g.R, oy - g.R, g.R, zlo, zhi) && has;
        nj = load_start(starts, base + zlo, has);
        nje = load_start(starts, base + zhi + 1, has);
    };
    request_row(live);
#pragma unroll 1
    for (;;) {
        if (!__any(j < je)) {   // every lane has finished its run: the next row
            if (round == n_rows) break;
            j = nj; je = nje;
            ++round;
            const bool wrap = oy + 1 == span;
            base += (round < n_rows) ? (wrap ? row_jump : g.np) : 0;
            oy = wrap ? 0 : oy + 1;
            ox += wrap ? 1 : 0;
            request_row(live && round < n_rows);
        }
        p3m_pool_vf cx, cy, cz;
        if constexpr (WIDE) {
            const int bb = j < je ? j : p;
#pragma unroll
            for (int u = 0; u < P3M_POOL_TRIP; ++u) {
                cx[u] = a.fx[static_cast<size_t>(bb) + u]; cy[u] = a.fy[static_cast<size_t>(bb) + u]; cz[u] = a.fz[static_cast<size_t>(bb) + u];
            }
        } else {
            const uint32_t off = j < je ? static_cast<uint32_t>(j) << 2 : 0xfffffff0u;
            cx = p3m_fvec_load<P3M_POOL_TRIP>::get(fx_rsrc, off);
            cy = p3m_fvec_load<P3M_POOL_TRIP>::get(fy_rsrc, off);
            cz = p3m_fvec_load<P3M_POOL_TRIP>::get(fz_rsrc, off);
        }
        const p3m_pool_vf dx = cx + nmx, dy = cy + nmy, dz = cz + nmz;
        const p3m_pool_vf r2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, dz * dz));
        const int left = je - j;        // candidates of this trip that belong to the run (may be <= 0)
        const unsigned entry0 = (static_cast<unsigned>(lane) << P3M_POOL_LANE_SHIFT) | static_cast<unsigned>(j);
#pragma unroll
        for (int u = 0; u < P3M_POOL_TRIP; ++u) {
            // two compares straight into lane masks, combined on the scalar side
            const bool in_run = u < left, near = !(r2[u] > bound);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(in_run) & __builtin_amdgcn_ballot_w64(near);
            // the survivors of this slot go to consecutive ring entries in lane order
            // (byte address: the rank among the survivors, scaled, on the scalar side's wrapped write position)
            constexpr int ES = WIDE ? 3 : 2;
            const unsigned rank = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(m >> 32),
                                                            __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(m), 0u));
            const unsigned at = ((rank << ES) + ((static_cast<unsigned>(wp) & (P3M_POOL_CAP - 1)) << ES)) & ((P3M_POOL_CAP - 1) << ES);
            Entry entry;
            if constexpr (WIDE) entry = make_uint2(static_cast<unsigned>(lane), static_cast<unsigned>(j + u));
            else entry = entry0 + static_cast<unsigned>(u);
            if (in_run & near) *reinterpret_cast<Entry*>(reinterpret_cast<char*>(ring) + at) = entry;
            wp += __popcll(m);
            if (u % 4 == 3)       // at most 127 left + 4 * 64 new entries in a ring of 512
                while (wp - rp >= 128) drain(std::integral_constant<int, 2>{}, std::true_type{}, 128);
        }
        j += P3M_POOL_TRIP;
    }
    while (wp - rp > 0) drain(std::integral_constant<int, 1>{}, std::false_type{}, wp - rp);
    __builtin_amdgcn_wave_barrier();

    if (live) {
        const int o = a.entries[p] - a.tgt_begin;
        a.ax[o] += tsum[0][threadIdx.x];
        a.ay[o] += tsum[1][threadIdx.x];
        a.az[o] += tsum[2][threadIdx.x];
    }
    if (a.accepted) {
        // every live target met itself once
        const long long h = static_cast<long long>(hits) - __popcll(__builtin_amdgcn_ballot_w64(live));
        if (lane == 0 && h) atomicAdd(accepted_stripe(a.accepted), static_cast<unsigned long long>(h));
    }
}

// ---------------------------------------------------------------------------------
// The same search with G lanes of a wavefront walking every row of ONE target together.
//
// Small particle sets cannot fill the GPU with a lane per target, and once they have clustered
// their work sits in a few thousand targets whose heaviest rows hold thousands of candidates:
// k_p3m_cells gives a target up to 16 lanes, but a lane owns whole rows, and the wavefront waits for
// the lane that got the row through the dense cell (32^3 at a = 0.9: 5.6 M accepted pairs in 0.2 ms,
// 28 G/s, where the arithmetic is 11 us of the GPU).  Here lane `sub` of a target's G lanes takes
// candidates sub, sub + G, ... of every row - consecutive lanes read consecutive 32-byte records - so
// the lanes of a target finish together whatever the rows hold.  No fp32 pre-filter: with one target
// per group the row is trimmed to the cells its cutoff sphere meets (row_window), after which more
// than half of the candidates are accepted, and the exact test on the fp64 record costs less than
// the filter's queue.  Every lane sums its own candidates in order and the G partial sums are added
// in a fixed tree, so with the members of a cell in a fixed order the result is reproducible.
// It takes the calls that k_p3m_cells would have taken with several lanes per target (small
// sets): measured against that kernel at its own best lane count, pair kernel alone, with 8 lanes
// per target (4 and 16 are within 10 % of it, 32 and 64 waste most lanes on short rows): the
// reference's shipped 32^3 run at a = 0.29 / 0.45 / 0.65 / 0.90 0.058 / 0.068 / 0.080 / 0.089 ms
// against 0.050 / 0.083 / 0.127 / 0.189 ms, a 64^3 run at a = 0.45 / 0.62 / 0.83 0.28 / 0.34 /
// 0.45 ms against 0.34 / 0.61 / 0.97 ms; and since it needs no particular order inside a cell
// unless reproducible sums are asked for, the binning before it is five launches shorter (0.056
// against 0.07-0.09 ms at 32^3), which pays for the one state where the kernel itself is slower.
// ---------------------------------------------------------------------------------
constexpr int P3M_ROWS_ABREAST = 2;   // rows of a target walked side by side (1, 2, 4: 0.061 / 0.058 / 0.058 ms on a
                                      // near-uniform 32^3 set, 0.089 / 0.091 / 0.099 ms on a clustered one)
constexpr int P3M_ROWS_LANES = 8;     // lanes per target when the kernel is chosen by itself
constexpr double P3M_ROWS_LARGE_MIN_OCCUPANCY = 1024.0; // clustering measure from which it also takes sets that fill the GPU: beyond
                                                        // anything measured (128^3 in blobs, measure 95 / 186 / 298 / 442: 6.97 / 11.4 /
                                                        // 8.07 / 11.3 ms against 6.19 / 11.0 / 7.61 / 10.9 with k_p3m_rows_pool)
constexpr double P3M_ROWS_POOL_MIN_OCCUPANCY = 2.7;     // ... and from which k_p3m_rows_pool takes them below that
constexpr int P3M_ROWS_LARGE_LANES = 4;                 // with one lane per target, and its lanes per target there

template <bool SOFT0, int G>
__global__ void __launch_bounds__(64) k_p3m_rows(const P3MArgs a) {
    static_assert(G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "lanes per target");
    constexpr int PER_WAVE = 64 / G;
    const int lane = threadIdx.x, sub = lane % G;
    const int t = static_cast<int>(blockIdx.x) * PER_WAVE + lane / G;
    const bool live = t < a.n_tgt;
    const int p = a.tlist[live ? t : a.n_tgt - 1];
    const double4 me = a.rec[p];
    const GhostGrid g = a.g;
    const PairConsts k = a.k;
    const int pcx = bin_coord(me.x, g.scale, g.nside) + g.R;
    const int pcy = bin_coord(me.y, g.scale, g.nside) + g.R;
    const int pcz = bin_coord(me.z, g.scale, g.nside) + g.R;
    const RowTrim trim = make_row_trim(me, pcx, pcy, pcz, g, k);
    const StartTable<true> starts = make_start_table<true>(a.cell_start, g);
    const uint32_t rec_bytes = static_cast<uint32_t>(a.n_entries) << 5;
    const __amdgpu_buffer_rsrc_t rec_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double4*>(a.rec), 0, rec_bytes, 0x00020000);
    double ax = 0.0, ay = 0.0, az = 0.0;
    unsigned long long hits = 0;
    const int span = 2 * g.R + 1;
    const int n_rows = span * span;
    // bounds of the part of row `row` this target can reach (none for a padding lane or past the last row)
    auto row_bounds = [&](int row, int& j, int& je) {
        const int ox = row / span, oy = row - ox * span;
        int zlo = 0, zhi = 0;
        const bool reach = row_window(trim, ox - g.R, oy - g.R, g.R, zlo, zhi);
        const bool has = live && row < n_rows && reach;
        const int base = ((pcx - g.R + ox) * g.np + (pcy - g.R + oy)) * g.np + pcz;
        j = load_start(starts, base + zlo, has);
        je = load_start(starts, base + zhi + 1, has);
    };
    auto candidate = [&](const double4& s, bool valid, int j) {
        const double dx = s.x - me.x, dy = s.y - me.y, dz = s.z - me.z;
        const double r2 = __builtin_fma(dx, dx, __builtin_fma(dy, dy, dz * dz));
        const bool accept = valid && j != p && !(r2 > k.cutoff_sq);   // the reference's cutoff test (src/particles.cpp:326)
        hits += __popcll(__builtin_amdgcn_ballot_w64(accept));        // the wavefront's count, kept on the scalar side
        if (accept) p3m_accepted<SOFT0>(dx, dy, dz, r2, s.w, k, ax, ay, az);
    };
    // ROWS rows are walked side by side, two candidates of each per trip: most rows of a small set
    // are a handful of candidates long, and one row at a time the kernel does little but wait for
    // the next row's bounds and records
    constexpr int ROWS = P3M_ROWS_ABREAST;
    int nj[ROWS], nje[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) row_bounds(r, nj[r], nje[r]);
#pragma unroll 1
    for (int row = 0; row < n_rows; row += ROWS) {
        int j[ROWS], je[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { j[r] = nj[r] + sub; je[r] = nje[r]; }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) row_bounds(row + ROWS + r, nj[r], nje[r]);   // in flight while these rows are walked
        for (;;) {
            bool any = false;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) any |= j[r] < je[r];
            if (!__any(any)) break;
            double4 s[ROWS][2];
            bool v[ROWS][2];
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    v[r][u] = j[r] + u * G < je[r];
                    s[r][u] = load_record_checked(rec_rsrc, v[r][u] ? static_cast<uint32_t>(j[r] + u * G) << 5 : 0xffffffe0u);
                }
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
#pragma unroll
                for (int u = 0; u < 2; ++u) candidate(s[r][u], v[r][u], j[r] + u * G);
                j[r] += 2 * G;
            }
        }
    }
    // the G lanes of a target sit next to each other: fixed-order tree sum
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
        ax += __shfl_xor(ax, d, 64);
        ay += __shfl_xor(ay, d, 64);
        az += __shfl_xor(az, d, 64);
    }
    if (live && sub == 0) {
        const int o = a.entries[p] - a.tgt_begin;
        a.ax[o] += ax; a.ay[o] += ay; a.az[o] += az;
    }
    if (a.accepted) {
        const unsigned long long h = hits;
        if (lane == 0 && h) atomicAdd(accepted_stripe(a.accepted), h);
    }
}

// ---------------------------------------------------------------------------------
// Row walk and pooled evaluation together.
//
// k_p3m_rows evaluates a candidate in the lane that tested it: where four candidates in ten pass
// the exact test, six lanes in ten sit out the 67 issue slots of the arithmetic.  Here the G lanes
// of a target walk its (trimmed) rows as there - lane `sub` takes four consecutive candidates at
// sub * 4, (sub + G) * 4, ... of a row, tested in single precision on the packed fp32 coordinates
// - and the survivors go into the wavefront's ring as (target slot, source) pairs, to be
// evaluated 128 at a time by whatever lane, the sums gathered per target in LDS with ds_add_f64:
// the drain of k_p3m_pool.  Not bitwise reproducible (colliding adds land in hardware order).
// ---------------------------------------------------------------------------------
template <bool SOFT0, int G>
__global__ void __launch_bounds__(64) k_p3m_rows_pool(const P3MArgs a) {
    static_assert(G == 2 || G == 4 || G == 8 || G == 16, "lanes per target");
    constexpr int PER_WAVE = 64 / G;
    typedef float vf4 __attribute__((ext_vector_type(4)));
    // ring entry = target slot << 27 | sorted position of the source (< 2^26): shifted left by five, to the
    // byte offset of the source's record, the slot is gone
    constexpr int SLOT_SHIFT = 27;
    struct Target { double2 xy; double z, pad; };      // one LDS address serves both reads
    __shared__ __align__(8) unsigned ring[P3M_POOL_CAP];
    __shared__ Target txyz[PER_WAVE];
    __shared__ double tsum[3][PER_WAVE];
    const int lane = threadIdx.x, sub = lane % G, slot = lane / G;
    const int t = static_cast<int>(blockIdx.x) * PER_WAVE + slot;
    const bool live = t < a.n_tgt;
    const int p = a.tlist[live ? t : a.n_tgt - 1];
    const double4 me = a.rec[p];
    const vf4 nmx = -a.fx[p], nmy = -a.fy[p], nmz = -a.fz[p];   // splats
    const GhostGrid g = a.g;
    const PairConsts k = a.k;
    const float bound = a.prefilter_sq;
    const int pcx = bin_coord(me.x, g.scale, g.nside) + g.R;
    const int pcy = bin_coord(me.y, g.scale, g.nside) + g.R;
    const int pcz = bin_coord(me.z, g.scale, g.nside) + g.R;
    if (sub == 0) {
        txyz[slot].xy = make_double2(me.x, me.y);
        txyz[slot].z = me.z;
        tsum[0][slot] = 0.0;
        tsum[1][slot] = 0.0;
        tsum[2][slot] = 0.0;
    }
    __builtin_amdgcn_wave_barrier();   // one wavefront: its LDS operations stay in order
    int rp = 0, wp = 0;                // ring read / write positions, unwrapped, the same in every lane
    unsigned long long hits = 0;                 // accepted pairs of the wavefront (the same in every lane)
    const uint32_t f_bytes = static_cast<uint32_t>(a.n_entries + P3M_POOL_TRIP) << 2;
    const __amdgpu_buffer_rsrc_t fx_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fx), 0, f_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t fy_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fy), 0, f_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t fz_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fz), 0, f_bytes, 0x00020000);
    const uint32_t rec_bytes = static_cast<uint32_t>(a.n_entries) << 5;
    const __amdgpu_buffer_rsrc_t rec_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double4*>(a.rec), 0, rec_bytes, 0x00020000);

    // one drain step: NB * 64 pairs from the front of the ring, NB per lane in flight together
    // FULL: the step takes NB * 64 entries, every lane has NB pairs.
    auto drain = [&](auto nb_tag, auto full_tag, int count) {
        constexpr int NB = decltype(nb_tag)::value;
        constexpr bool FULL = decltype(full_tag)::value;
        unsigned ts[NB], en[NB];      // target slot, ring entry
        bool valid[NB];
        double4 s[NB];
        double2 gxy[NB];
        double gz[NB];
        // a lane takes NB consecutive entries: a target's survivors lie together in the ring, and spread
        // over the NB evaluations fewer of one evaluation's adds meet at the same sum
        if constexpr (FULL && NB == 2) {   // full steps leave rp even: one 8-byte read, which does not wrap
            const uint2 both = *reinterpret_cast<const uint2*>(&ring[(rp + 2 * lane) & (P3M_POOL_CAP - 1)]);
            en[0] = both.x;
            en[1] = both.y;
        } else {
#pragma unroll
            for (int u = 0; u < NB; ++u) en[u] = ring[(rp + lane * NB + u) & (P3M_POOL_CAP - 1)];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            valid[u] = FULL || lane * NB + u < count;
            ts[u] = valid[u] ? en[u] >> SLOT_SHIFT : static_cast<unsigned>(slot);
            gxy[u] = txyz[ts[u]].xy;
            gz[u] = txyz[ts[u]].z;
            s[u] = load_record_checked(rec_rsrc, valid[u] ? en[u] << 5 : 0xffffffe0u);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const double dx = s[u].x - gxy[u].x, dy = s[u].y - gxy[u].y, dz = s[u].z - gz[u];
            const double r2 = __builtin_fma(dx, dx, __builtin_fma(dy, dy, dz * dz));
            const bool accept = valid[u] && !(r2 > k.cutoff_sq);   // the reference's cutoff test (src/particles.cpp:326)
            hits += __popcll(__builtin_amdgcn_ballot_w64(accept));   // counted on the scalar side
            if (accept) {
                double cx = 0.0, cy = 0.0, cz = 0.0;
                p3m_accepted<SOFT0>(dx, dy, dz, r2, s[u].w, k, cx, cy, cz);
                __hip_atomic_fetch_add(&tsum[0][ts[u]], cx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&tsum[1][ts[u]], cy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&tsum[2][ts[u]], cz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        rp += FULL ? NB * 64 : min(count, NB * 64);
    };

    const int span = 2 * g.R + 1;
    const int n_rows = span * span;
    const RowTrim trim = make_row_trim(me, pcx, pcy, pcz, g, k);
    const StartTable<true> starts = make_start_table<true>(a.cell_start, g);
    int nj = 0, nje = 0;
    auto request_row = [&](int row) {
        const int ox = row / span, oy = row - ox * span;
        int zlo = 0, zhi = 0;
        const bool reach = row_window(trim, ox - g.R, oy - g.R, g.R, zlo, zhi);
        const bool has = live && row < n_rows && reach;
        const int base = ((pcx - g.R + ox) * g.np + (pcy - g.R + oy)) * g.np + pcz;
        nj = load_start(starts, base + zlo, has);
        nje = load_start(starts, base + zhi + 1, has);
    };
    request_row(0);
    const unsigned tag = static_cast<unsigned>(slot) << SLOT_SHIFT;
#pragma unroll 1
    for (int row = 0; row < n_rows; ++row) {
        int j = nj + 4 * sub;
        const int je = nje;
        request_row(row + 1);          // in flight while this row is walked
#pragma unroll 1
        while (__any(j < je)) {
            const uint32_t off = j < je ? static_cast<uint32_t>(j) << 2 : 0xfffffff0u;
            const vf4 cx = p3m_fvec_load<4>::get(fx_rsrc, off);
            const vf4 cy = p3m_fvec_load<4>::get(fy_rsrc, off);
            const vf4 cz = p3m_fvec_load<4>::get(fz_rsrc, off);
            const vf4 dx = cx + nmx, dy = cy + nmy, dz = cz + nmz;
            const vf4 r2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, dz * dz));
            const int left = je - j;
            const unsigned entry0 = tag | static_cast<unsigned>(j);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in_run = u < left, near = !(r2[u] > bound);
                const unsigned long long m = __builtin_amdgcn_ballot_w64(in_run) & __builtin_amdgcn_ballot_w64(near);
                // byte address in the ring: the rank among this slot's survivors, times four, on the scalar
                // side's wrapped write position
                const unsigned rank = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(m >> 32),
                                                                __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(m), 0u));
                const unsigned at = ((rank << 2) + ((static_cast<unsigned>(wp) & (P3M_POOL_CAP - 1)) << 2)) & (4 * P3M_POOL_CAP - 4);
                if (in_run & near)
                    *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ring) + at) = entry0 + static_cast<unsigned>(u);
                wp += __popcll(m);
            }
            // at most 127 left + 4 * 64 new entries in a ring of 512
            while (wp - rp >= 128) drain(std::integral_constant<int, 2>{}, std::true_type{}, 128);
            j += 4 * G;
        }
    }
    while (wp - rp > 0) drain(std::integral_constant<int, 1>{}, std::false_type{}, wp - rp);
    __builtin_amdgcn_wave_barrier();
    if (live && sub == 0) {
        const int o = a.entries[p] - a.tgt_begin;
        a.ax[o] += tsum[0][slot];
        a.ay[o] += tsum[1][slot];
        a.az[o] += tsum[2][slot];
    }
    if (a.accepted) {
        // every live target met itself once
        const long long h = static_cast<long long>(hits) - __popcll(__builtin_amdgcn_ballot_w64(live && sub == 0));
        if (lane == 0 && h) atomicAdd(accepted_stripe(a.accepted), static_cast<unsigned long long>(h));
    }
}

// ---------------------------------------------------------------------------------
// Every unordered pair once (Newton's third law) for the P3M short range.
//
// The reference evaluates (i, j) and (j, i) separately (src/particles.cpp:270-365); their
// displacements are exact negatives and the scalar factor S (f_pp - f_pm) is the same, so one
// evaluation serves both: a_i += G m_j f d, a_j -= G m_i f d.  That halves the candidates a
// target has to test as well as the accepted pairs it has to evaluate: a target only looks
// FORWARD, at the rows (ox, oy) of its neighbourhood that follow its own in (x, y) order and at
// the entries behind itself in its own row.  With ghost images every unordered pair of particles
// is then met by exactly one of its two members (the image of a forward offset is a backward one).
//
// The reaction a_j has to be added to a particle another lane, wavefront or workgroup may be
// adding to as well.  A workgroup therefore owns a BRICK of bx x by x bz (SymBrick) search
// cells: its targets are the brick's real entries, and every source they can meet lies in the
// region [x0, x0 + BX + R) x [y0 - R, y0 + BY + R) x [z0 - R, z0 + BZ + R), a few dozen columns
// whose z segments are contiguous runs of the sorted records.  One fp64 sum triple per region
// entry lives in LDS; lanes add reactions there with ds_add_f64, a target adds its own sums to
// its own entry at the end, and when the brick is done every non-zero triple is added to the
// output with global fp64 atomics, once per region entry instead of once per pair (about 11
// atomics per target instead of 107).  Not bitwise reproducible; ordinary device memory only.
// ---------------------------------------------------------------------------------
constexpr int SYM_WG = 512;         // 8 wavefronts per brick; two bricks share a CU's LDS, so one's set-up and
                                    // flush run beside the other's pair loop
// A brick's extent in search cells: chosen per call (sym_brick) so that a brick holds about 440 targets at
// the density the search grid was laid out for, whatever the search radius.
struct SymBrick { int bx, by, bz; };
constexpr int SYM_MAXPIECE = 36;    // target columns of a brick (bx * by) the tables are sized for
constexpr int SYM_CAP = 1920;       // region entries whose sums fit in LDS (45 KiB)
constexpr int SYM_DU = 4;           // record fetches in flight per lane while draining (2 / 3 / 4 / 6: 4.79 / 4.47 / 4.39 / 4.37 ms at 256^3)
constexpr int SYM_MAXR = 3;         // search radius in cells the tables are sized for
constexpr int SYM_RETRY_CALLS = 16; // calls between two tries of the pre-pass once a brick region did not fit
constexpr int SYM_MAXCOL = 128;     // region columns ((bx + R) * (by + 2 R)) the tables are sized for
constexpr int SYM_LIST = 16;        // list entries per lane (one per trip with a survivor)
constexpr int SYM_JBITS = 25;       // (sorted positions the launch logic lets this kernel see: below 2^25)
constexpr int SYM_HALF = 8;         // candidates behind one list entry (its mask)
constexpr int SYM_HALVES = 2;       // list entries a trip of the filter can leave: 16 candidates per trip cover most
                                    // windows in one trip (1.2 trips per row and wavefront at 256^3 against 2.0 with 8)
constexpr int SYM_TRIP = SYM_HALF * SYM_HALVES;   // candidates a lane tests per trip of the filter
static_assert(SYM_HALF == 8 && SYM_TRIP <= 16 && SYM_LIST > SYM_HALVES && SYM_MAXCOL <= 128 && SYM_CAP <= 2048 && SYM_LIST >= 2, "list entry layout of k_p3m_sym");
static_assert(SYM_MAXPIECE <= 64, "one wavefront sums the piece lengths");

// 2R x 2R x 7R cells: the same physical brick (1 x 1 x 3.5 cutoff radii) for every search radius R
__host__ __device__ inline SymBrick sym_brick(int R) { return SymBrick{2 * R, 2 * R, 7 * R}; }
__host__ __device__ inline bool sym_brick_fits(const SymBrick& b, int R) {
    return R <= SYM_MAXR && b.bx * b.by <= SYM_MAXPIECE && (b.bx + R) * (b.by + 2 * R) <= SYM_MAXCOL;
}

// S (f_pp - f_pm) of src/particles.cpp:330-355 for r^2 <= cutoff_sq (without G m and d).
template <bool SOFT0>
__device__ __forceinline__ double p3m_factor(double r2, const PairConsts& k) {
    auto cube = [&](double q) {
        const double y = __builtin_amdgcn_rsq(q);
        const double y2 = y * y;
        const double h = __builtin_fma(-q, y2, 1.0);
        const double y3 = y * y2;
        return __builtin_fma(y3, h * __builtin_fma(k.c1875, h, k.c15), y3);
    };
    const double q1 = r2 + k.eps2;
    double f1 = cube(q1);
    if constexpr (SOFT0) f1 = (q1 > 0.0) ? f1 : 0.0;
    double f = f1 - cube(r2 + k.eps_pm2);
    if (r2 > k.r_switch_sq) {
        const double dist = r2 * inv_sqrt_refined(r2);
        const double t = (dist - k.r_switch) * k.inv_width;
        f *= __builtin_fma(t * t, __builtin_fma(2.0, t, -3.0), 1.0);
    }
    return f;
}

// Largest number of region entries over all bricks: the launch falls back to k_p3m_cells when
// one does not fit SYM_CAP (a clustered set).
__global__ void k_sym_region_max(GhostGrid g, const int* __restrict__ cell_start, SymBrick bk, int nbx, int nby, int nbz,
                                 unsigned long long* __restrict__ out) {
    // one thread per brick, bz fastest: neighbouring threads read neighbouring words of the same columns
    const int brick = blockIdx.x * blockDim.x + threadIdx.x;
    int total = 0;
    if (brick < nbx * nby * nbz) {
        const int bz = brick % nbz, by = (brick / nbz) % nby, bx = brick / (nbz * nby);
        const int x0 = g.R + bx * bk.bx, y0 = g.R + by * bk.by, z0 = g.R + bz * bk.bz;
        const int zlo = z0 - g.R, zhi = min(z0 + bk.bz, g.R + g.nside) + g.R;
        for (int rx = 0; rx < bk.bx + g.R; ++rx)
            for (int ry = 0; ry < bk.by + 2 * g.R; ++ry) {
                const int px = x0 + rx, py = y0 - g.R + ry;
                if (px >= g.np || py >= g.np) continue;
                const int col = (px * g.np + py) * g.np;
                total += cell_start[col + zhi] - cell_start[col + zlo];
            }
    }
    for (int d = 1; d < 64; d <<= 1) total = max(total, __shfl_xor(total, d, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, static_cast<unsigned long long>(total));
}

template <bool SOFT0>
__global__ void __launch_bounds__(SYM_WG, 4) k_p3m_sym(const P3MArgs a, SymBrick bk, int nbx, int nby, int nbz) {
    __shared__ double acc[3][SYM_CAP];                     // the reaction sums of every region entry
    __shared__ int list[SYM_LIST][SYM_WG];                 // per lane: the filter's survivors, one entry per trip
    __shared__ int col_off[SYM_MAXCOL + 1];                // first LDS slot of every region column
    __shared__ int col_delta[128];             // LDS slot = sorted position + col_delta[column] (SYM_MAXCOL used; a list entry's 7-bit column field can always be looked up)
    __shared__ int piece_off[SYM_MAXPIECE + 1], piece_start[SYM_MAXPIECE];   // the brick's targets, column by column
    const GhostGrid g = a.g;
    const PairConsts k = a.k;
    const int b = static_cast<int>(blockIdx.x);      // XCD-aware order of the bricks: see k_p3m_cells
    const int group = b / (8 * P3M_XCD_RUN);
    const int in_group = b - group * (8 * P3M_XCD_RUN);
    const bool whole = (group + 1) * (8 * P3M_XCD_RUN) <= static_cast<int>(gridDim.x);
    const int brick = whole ? group * (8 * P3M_XCD_RUN) + (in_group % 8) * P3M_XCD_RUN + in_group / 8 : b;
    const int bz = brick % nbz, by = (brick / nbz) % nby, bx = brick / (nbz * nby);
    const int x0 = g.R + bx * bk.bx, y0 = g.R + by * bk.by, z0 = g.R + bz * bk.bz;
    const int z1 = min(z0 + bk.bz, g.R + g.nside);       // the brick's real cells end here
    const int CY = bk.by + 2 * g.R, ncols = (bk.bx + g.R) * CY, npieces = bk.bx * bk.by;
    const int tid = threadIdx.x;

    // ---- the region's columns and the brick's target pieces
    if (tid < ncols) {
        const int rx = tid / CY, ry = tid - rx * CY;
        const int px = x0 + rx, py = y0 - g.R + ry;
        int first = 0, count = 0;
        if (px < g.np && py < g.np) {
            const int col = (px * g.np + py) * g.np;
            first = a.cell_start[col + z0 - g.R];
            count = a.cell_start[col + z1 + g.R] - first;
        }
        col_delta[tid] = first;      // for now: the column's first sorted position
        col_off[tid + 1] = count;    // for now: its length
    }
    if (tid < npieces) {
        const int rx = tid / bk.by, ry = tid - rx * bk.by;
        const int px = x0 + rx, py = y0 + ry;
        int first = 0, count = 0;
        if (px < g.R + g.nside && py < g.R + g.nside) {
            const int col = (px * g.np + py) * g.np;
            first = a.cell_start[col + z0];
            count = a.cell_start[col + z1] - first;
        }
        piece_start[tid] = first;
        piece_off[tid + 1] = count;
    }
    __syncthreads();
    if (tid < 64) {               // first wavefront: running sums of the column lengths, 64 columns at a time
        int carry = 0;
        for (int c0 = 0; c0 < ncols; c0 += 64) {
            const int c = c0 + tid;
            const int count = c < ncols ? col_off[c + 1] : 0;
            int incl = count;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d, 64);
                incl += tid >= d ? up : 0;
            }
            if (c < ncols) {
                col_delta[c] = carry + incl - count - col_delta[c];
                col_off[c + 1] = carry + incl;
            }
            carry += __shfl(incl, 63, 64);
        }
        if (tid == 0) col_off[0] = 0;
    } else if (tid < 128) {       // second wavefront: the same for the target pieces
        const int l = tid - 64;
        const int count = l < npieces ? piece_off[l + 1] : 0;
        int incl = count;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d, 64);
            incl += l >= d ? up : 0;
        }
        if (l < npieces) piece_off[l + 1] = incl;
        if (l == 0) piece_off[0] = 0;
    }
    __syncthreads();
    const int n_region = col_off[ncols];
    const int n_mine = piece_off[npieces];
    for (int e = tid; e < n_region; e += SYM_WG) {
        acc[0][e] = 0.0;
        acc[1][e] = 0.0;
        acc[2][e] = 0.0;
    }
    __syncthreads();

    // quantised candidate positions: x | y << 16 per entry, z in a half word per entry (SYM_TRIP entries of
    // padding behind both, so a trip that starts at a valid entry never leaves the buffers)
    const uint32_t qxy_bytes = static_cast<uint32_t>(a.n_entries + SYM_TRIP) << 2, qz_bytes = static_cast<uint32_t>(a.n_entries + SYM_TRIP) << 1;
    const __amdgpu_buffer_rsrc_t qxy_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.qxy), 0, qxy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t qz_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.qz), 0, qz_bytes, 0x00020000);
    const int nbound = -(a.qbound + 1);       // r_q^2 + nbound < 0  <=>  r_q^2 <= qbound
    const StartTable<true> starts = make_start_table<true>(a.cell_start, g);
    const uint32_t rec_bytes = static_cast<uint32_t>(a.n_entries) << 5;
    const __amdgpu_buffer_rsrc_t rec_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double4*>(a.rec), 0, rec_bytes, 0x00020000);
    const int span = 2 * g.R + 1;
    const int n_rows = g.R * span + g.R + 1;     // the own row, then every row after it in (x, y) order
    unsigned long long hits = 0;
    constexpr int DU = SYM_DU;
#ifdef NBPP_SYM_STATS
    unsigned long long dbg_trips = 0, dbg_rows = 0, dbg_steps = 0, dbg_live = 0, dbg_cand = 0, dbg_surv = 0;
#endif

#pragma unroll 1
    for (int chunk = 0; chunk * SYM_WG < n_mine; ++chunk) {
        const int t = chunk * SYM_WG + tid;
        const bool live = t < n_mine;
        if (!__any(live)) continue;          // a wavefront without targets in this chunk
        int piece = 0;
        for (int c = 1; c < npieces; ++c) piece += (live && t >= piece_off[c]) ? 1 : 0;
        const int p = live ? piece_start[piece] + (t - piece_off[piece]) : 0;
        const double4 me = a.rec[p];
        const int rcx = piece / bk.by, rcy = piece - rcx * bk.by + g.R;   // own column inside the region
        const int own = live ? p + col_delta[rcx * CY + rcy] : 0;           // own LDS slot
        typedef short s2 __attribute__((ext_vector_type(2)));
        const s2 txy = __builtin_bit_cast(s2, a.qxy[p]);                    // own quantised x | y
        const unsigned tz1 = a.qz[p];
        const s2 tzz = __builtin_bit_cast(s2, tz1 | (tz1 << 16));           // own quantised z, twice
        const int pcx = x0 + rcx, pcy = y0 + rcy - g.R;
        const int pcz = bin_coord(me.z, g.scale, g.nside) + g.R;
        const RowTrim trim = make_row_trim(me, pcx, pcy, pcz, g, k);
        double ax = 0.0, ay = 0.0, az = 0.0;
        int cnt = 0;                 // entries in this lane's list (at most SYM_LIST)

        // The filter leaves, per trip of SYM_TRIP candidates with at least one survivor, ONE list entry:
        //   bits 0-7   which of the trip's candidates survived (candidate u at bit 7 - u),
        //   bits 8-14  the region column of the row,
        //   bits 15-25 the LDS slot of the trip's first candidate (sorted position + col_delta[column]).
        // Evaluating pops the bits: every lane walks its own list at its own pace, a pair per step, until
        // all lists of the wavefront are empty.  The next entry and the col_delta of its column (slot
        // - col_delta = the source's sorted position, where its fp64 record lies) are fetched from LDS
        // one entry ahead of their use, so a lane that moves to its next entry never waits for LDS.
        // (Letting an evaluation stop early, with the rest waiting in a ring for the next one, was measured:
        // the lists rarely fill before the rows are used up, so there is one evaluation per wavefront
        // anyway and its length is that of the lane with the most pairs; no gain.)
        auto drain_all = [&]() {
            int rd = 0;                                       // entries that have entered the pipeline
            unsigned e_next = static_cast<unsigned>(list[0][tid]);
            unsigned e_after = static_cast<unsigned>(list[1][tid]);
            int d_next = col_delta[(e_next >> 8) & 127u];
            unsigned mask = 0u;
            int base7 = 0, delta = 0;
#pragma unroll 1
            while (__any(mask != 0u || rd < cnt)) {
                double4 s[DU];
                bool valid[DU];
                int slot[DU];
#pragma unroll
                for (int u = 0; u < DU; ++u) {
                    if (mask == 0u && rd < cnt) {             // on to the next entry (never an empty one)
                        mask = e_next & 255u;
                        base7 = static_cast<int>(e_next >> 15) + 7;
                        delta = d_next;
                        e_next = e_after;
                        d_next = col_delta[(e_after >> 8) & 127u];
                        ++rd;
                        e_after = static_cast<unsigned>(list[min(rd + 1, SYM_LIST - 1)][tid]);
                    }
                    valid[u] = mask != 0u;
                    const int bit = __builtin_ctz(mask | 0x100u);
                    mask &= mask - 1u;
                    slot[u] = base7 - bit;
                    const unsigned jj = static_cast<unsigned>(slot[u] - delta);
                    s[u] = load_record_checked(rec_rsrc, valid[u] ? jj << 5 : 0xffffffe0u);
#ifdef NBPP_SYM_STATS
                    dbg_steps += 1;
                    dbg_live += __popcll(__builtin_amdgcn_ballot_w64(valid[u]));
#endif
                }
#pragma unroll
                for (int u = 0; u < DU; ++u) {
                    const double dx = s[u].x - me.x, dy = s[u].y - me.y, dz = s[u].z - me.z;
                    const double r2 = __builtin_fma(dx, dx, __builtin_fma(dy, dy, dz * dz));
                    const bool accept = valid[u] && !(r2 > k.cutoff_sq);     // src/particles.cpp:326
                    hits += __popcll(__builtin_amdgcn_ballot_w64(accept));   // the wavefront's count, kept on the scalar side
                    if (accept) {
                        const double f = p3m_factor<SOFT0>(r2, k);
                        const double fi = s[u].w * f;          // a_i += G m_j f d
                        ax = __builtin_fma(fi, dx, ax);
                        ay = __builtin_fma(fi, dy, ay);
                        az = __builtin_fma(fi, dz, az);
                        const double fj = -me.w * f;           // a_j -= G m_i f d
                        __hip_atomic_fetch_add(&acc[0][slot[u]], fj * dx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(&acc[1][slot[u]], fj * dy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(&acc[2][slot[u]], fj * dz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            cnt = 0;
        };

        // row r: 0 the own row (entries behind p only), 1..R the rows (0, 1..R), then (1..R, -R..R);
        // called with r = 0, 1, 2, ...: (ox, oy) step along
        int ox = 0, oy = -1;
        auto row_bounds = [&](int r, int& first, int& last, int& tg, int& dl) {
            ++oy;
            if (oy > g.R) { oy = -g.R; ++ox; }
            int zlo, zhi;                       // the part of the row this particle can reach
            const bool has = row_window(trim, ox, oy, g.R, zlo, zhi) && live && r < n_rows;
            const int base = ((pcx + ox) * g.np + (pcy + oy)) * g.np + pcz;
            first = load_start(starts, base + zlo, has);
            last = load_start(starts, base + zhi + 1, has);
            if (r == 0) first = has ? p + 1 : 0;
            const int col = min((rcx + ox) * CY + (rcy + oy), SYM_MAXCOL - 1);   // (lanes without a row: any valid column)
            tg = col << 8;
            dl = col_delta[col];
        };
        int round = 0;
        int j = 0, jb = 0, je = 0, tag = 0, dlt = 0;      // the run is [jb, je); trips start at even positions
        int nj, nje, ntag, ndlt;
        row_bounds(0, nj, nje, ntag, ndlt);
        // moves (j, jb, je, tag, dlt) to the next trip of SYM_TRIP candidates; false once the rows are used up
        auto next_trip = [&]() -> bool {
            j += SYM_TRIP;
            if (!__any(j < je)) {
                if (round == n_rows) return false;
                jb = nj; je = nje; tag = ntag; dlt = ndlt;
                j = jb & ~1;                    // an even position: the half words of z then start on a word
                ++round;
#ifdef NBPP_SYM_STATS
                dbg_rows += 1;
#endif
                row_bounds(round, nj, nje, ntag, ndlt);
            }
            return true;
        };
#pragma unroll 1
        while (next_trip()) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
#ifdef NBPP_SYM_STATS
            dbg_trips += 1;
#endif
            if (j < je) {                       // (an exhausted lane fetches and tests nothing)
                u4 xy[2 * SYM_HALVES], zz[SYM_HALVES];
#pragma unroll
                for (int h = 0; h < SYM_HALVES; ++h) {
                    xy[2 * h] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(qxy_rsrc, (static_cast<uint32_t>(j) << 2) + 32u * h, 0, 0));
                    xy[2 * h + 1] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(qxy_rsrc, (static_cast<uint32_t>(j) << 2) + 32u * h + 16u, 0, 0));
                    zz[h] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(qz_rsrc, (static_cast<uint32_t>(j) << 1) + 16u * h, 0, 0));
                }
                // r_q^2 + nbound per candidate in exact integer arithmetic: packed 16-bit differences, x^2 + y^2
                // by v_dot2, z^2 by a 24-bit multiply-add; its sign bit is set <=> the candidate may lie inside
                // the cutoff.  The sign bits are shifted into one word, first candidate first (v_alignbit).
                unsigned bits = 0u;
#pragma unroll
                for (int u = 0; u < SYM_TRIP; ++u) {
                    // (vector elements go through scalars: a bit cast of the element itself reads element 0)
                    const unsigned cxy = xy[u >> 2][u & 3];
                    const unsigned czz = zz[u >> 3][(u >> 1) & 3];
                    const s2 dxy = __builtin_bit_cast(s2, cxy) - txy;
                    const s2 dzz = __builtin_bit_cast(s2, czz) - tzz;
                    const int dz = (u & 1) ? dzz.y : dzz.x;
                    const int rq = __builtin_amdgcn_sdot2(dxy, dxy, dz * dz + nbound, false);
                    bits = __builtin_amdgcn_alignbit(bits, static_cast<unsigned>(rq), 31);
                }
                // only positions jb <= j + u < je belong to the run
                constexpr unsigned FULL = (1u << SYM_TRIP) - 1u;
                const int lead = max(jb - j, 0), left = min(je - j, SYM_TRIP);
                const unsigned mask = bits & (FULL >> lead) & ((FULL << SYM_TRIP) >> left);
#ifdef NBPP_SYM_STATS
                for (int u = 0; u < SYM_TRIP; ++u) dbg_cand += __popcll(__builtin_amdgcn_ballot_w64(u >= lead && u < left));
                for (int u = 0; u < SYM_TRIP; ++u) dbg_surv += __popcll(__builtin_amdgcn_ballot_w64((mask >> (SYM_TRIP - 1 - u)) & 1u));
#endif
                // one entry per group of eight candidates with a survivor (cnt <= SYM_LIST - SYM_HALVES here:
                // a list without room for a whole trip is drained at once, below)
#pragma unroll
                for (int h = 0; h < SYM_HALVES; ++h) {
                    const unsigned m8 = (mask >> (SYM_HALF * (SYM_HALVES - 1 - h))) & 255u;
                    list[cnt][tid] = static_cast<int>((static_cast<unsigned>(j + SYM_HALF * h + dlt) << 15) | static_cast<unsigned>(tag) | m8);
                    cnt += m8 != 0u ? 1 : 0;
                }
            }
            if (__any(cnt > SYM_LIST - SYM_HALVES)) drain_all();      // some lane's list has no room for another trip
        }
        drain_all();
        if (live) {   // the target's own sums join what others add to it
            __hip_atomic_fetch_add(&acc[0][own], ax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&acc[1][own], ay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&acc[2][own], az, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();

    // ---- every region entry that received something goes out once (FL entries per lane in
    // flight: the look-up of the particle's index is a dependent global load)
    constexpr int FL = 4;
    for (int e0 = tid; e0 < n_region; e0 += FL * SYM_WG) {
        int o[FL];
        bool any[FL];
#pragma unroll
        for (int u = 0; u < FL; ++u) {
            const int e = e0 + u * SYM_WG;
            any[u] = e < n_region && (acc[0][e] != 0.0 || acc[1][e] != 0.0 || acc[2][e] != 0.0);
            int lo = 0, hi = ncols;                  // column of entry e: col_off[lo] <= e < col_off[lo + 1]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (col_off[mid] <= e) lo = mid; else hi = mid;
            }
            o[u] = any[u] ? a.entries[e - col_delta[lo]] - a.tgt_begin : 0;
        }
#pragma unroll
        for (int u = 0; u < FL; ++u) {
            if (!any[u]) continue;
            const int e = e0 + u * SYM_WG;
            unsafeAtomicAdd(&a.ax[o[u]], acc[0][e]);
            unsafeAtomicAdd(&a.ay[o[u]], acc[1][e]);
            unsafeAtomicAdd(&a.az[o[u]], acc[2][e]);
        }
    }
    if (a.accepted) {
        const unsigned long long h = 2ull * hits;    // both members of every pair
        if ((tid & 63) == 0 && h) atomicAdd(accepted_stripe(a.accepted), h);
#ifdef NBPP_SYM_STATS
        if ((tid & 63) == 0) {
            unsigned long long* d = accepted_stripe(a.accepted);
            atomicAdd(d + 1, dbg_trips); atomicAdd(d + 2, dbg_rows); atomicAdd(d + 3, dbg_steps);
            atomicAdd(d + 4, dbg_live); atomicAdd(d + 5, dbg_cand); atomicAdd(d + 6, dbg_surv); atomicAdd(d + 7, 1ull);
        }
#endif
    }
}

// Sum the per-slice partials in slice order (deterministic) and accumulate into the output.
__global__ void k_reduce_partials(int n_tgt, int n_tgt_pad, int n_split,
                                  const double* __restrict__ px, const double* __restrict__ py,
                                  const double* __restrict__ pz, const int* __restrict__ tperm,
                                  double* __restrict__ ax, double* __restrict__ ay,
                                  double* __restrict__ az) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_tgt) return;
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int s = 0; s < n_split; ++s) {
        const size_t o = static_cast<size_t>(s) * n_tgt_pad + idx;
        sx += px[o]; sy += py[o]; sz += pz[o];
    }
    const int o = tperm[idx];
    ax[o] += sx; ay[o] += sy; az[o] += sz;
}

// ---------------------------------------------------------------------------------
// Caller-contract kernels: HBM-bandwidth-bound streams.
// ---------------------------------------------------------------------------------
// v += a*(dt/a^3) - v*(2 H dt)                                   src/integrator.cpp:168-181
__global__ void k_kick(size_t n, double* __restrict__ vx, double* __restrict__ vy,
                       double* __restrict__ vz, const double* __restrict__ ax,
                       const double* __restrict__ ay, const double* __restrict__ az,
                       double dt_over_a3, double dt_2H) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        // two roundings per product and one per add/sub, like the reference's statement
        vx[i] = vx[i] + (__dmul_rn(ax[i], dt_over_a3) - __dmul_rn(vx[i], dt_2H));
        vy[i] = vy[i] + (__dmul_rn(ay[i], dt_over_a3) - __dmul_rn(vy[i], dt_2H));
        vz[i] = vz[i] + (__dmul_rn(az[i], dt_over_a3) - __dmul_rn(vz[i], dt_2H));
    }
}

// a += b                                                          src/particles.cpp:362-364
template <bool CLEAR>   // CLEAR: b is left holding zeros (the sums of the next evaluation start from them)
__global__ void k_add(size_t n, double* __restrict__ ax, double* __restrict__ ay, double* __restrict__ az,
                      double* __restrict__ bx, double* __restrict__ by, double* __restrict__ bz) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        ax[i] = __dadd_rn(ax[i], bx[i]);
        ay[i] = __dadd_rn(ay[i], by[i]);
        az[i] = __dadd_rn(az[i], bz[i]);
        if constexpr (CLEAR) { bx[i] = 0.0; by[i] = 0.0; bz[i] = 0.0; }
    }
}

// x += v*dt ; x -= L*floor(x/L)                                  src/integrator.cpp:183-198
__global__ void k_drift(size_t n, double* __restrict__ x, double* __restrict__ y,
                        double* __restrict__ z, const double* __restrict__ vx,
                        const double* __restrict__ vy, const double* __restrict__ vz, double dt,
                        double L, double inv_L) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double nx = __dadd_rn(x[i], __dmul_rn(vx[i], dt));
        const double ny = __dadd_rn(y[i], __dmul_rn(vy[i], dt));
        const double nz = __dadd_rn(z[i], __dmul_rn(vz[i], dt));
        x[i] = __dsub_rn(nx, __dmul_rn(L, floor(__dmul_rn(nx, inv_L))));
        y[i] = __dsub_rn(ny, __dmul_rn(L, floor(__dmul_rn(ny, inv_L))));
        z[i] = __dsub_rn(nz, __dmul_rn(L, floor(__dmul_rn(nz, inv_L))));
    }
}

// k_kick then k_drift in one pass over the particles: the same operations on every element in
// the same order, so the results are those of the two kernels bit for bit.
__global__ void k_kick_drift(size_t n, double* __restrict__ x, double* __restrict__ y, double* __restrict__ z,
                             double* __restrict__ vx, double* __restrict__ vy, double* __restrict__ vz,
                             const double* __restrict__ ax, const double* __restrict__ ay,
                             const double* __restrict__ az, double dt_over_a3, double dt_2H, double dt,
                             double L, double inv_L) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double ux = vx[i] + (__dmul_rn(ax[i], dt_over_a3) - __dmul_rn(vx[i], dt_2H));
        const double uy = vy[i] + (__dmul_rn(ay[i], dt_over_a3) - __dmul_rn(vy[i], dt_2H));
        const double uz = vz[i] + (__dmul_rn(az[i], dt_over_a3) - __dmul_rn(vz[i], dt_2H));
        vx[i] = ux; vy[i] = uy; vz[i] = uz;
        const double nx = __dadd_rn(x[i], __dmul_rn(ux, dt));
        const double ny = __dadd_rn(y[i], __dmul_rn(uy, dt));
        const double nz = __dadd_rn(z[i], __dmul_rn(uz, dt));
        x[i] = __dsub_rn(nx, __dmul_rn(L, floor(__dmul_rn(nx, inv_L))));
        y[i] = __dsub_rn(ny, __dmul_rn(L, floor(__dmul_rn(ny, inv_L))));
        z[i] = __dsub_rn(nz, __dmul_rn(L, floor(__dmul_rn(nz, inv_L))));
    }
}

// k_add<true> then k_kick in one pass: a += b, b = 0, v kicked with the new a.  WITH_MAX: the
// per-block maxima of k_max_accel_sq over the new a as well (the same roundings and the same
// strict compares, so the reduced value is the separate kernel's bit for bit).
template <bool WITH_MAX>
__global__ void __launch_bounds__(WG) k_add_clear_kick(size_t n, double* __restrict__ ax, double* __restrict__ ay,
                                                       double* __restrict__ az, double* __restrict__ bx,
                                                       double* __restrict__ by, double* __restrict__ bz,
                                                       double* __restrict__ vx, double* __restrict__ vy,
                                                       double* __restrict__ vz, double dt_over_a3, double dt_2H,
                                                       double* __restrict__ block_max) {
    __shared__ double lds[WG / 64];
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    double m = 0.0;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double sx = __dadd_rn(ax[i], bx[i]), sy = __dadd_rn(ay[i], by[i]), sz = __dadd_rn(az[i], bz[i]);
        ax[i] = sx; ay[i] = sy; az[i] = sz;
        bx[i] = 0.0; by[i] = 0.0; bz[i] = 0.0;
        vx[i] = vx[i] + (__dmul_rn(sx, dt_over_a3) - __dmul_rn(vx[i], dt_2H));
        vy[i] = vy[i] + (__dmul_rn(sy, dt_over_a3) - __dmul_rn(vy[i], dt_2H));
        vz[i] = vz[i] + (__dmul_rn(sz, dt_over_a3) - __dmul_rn(vz[i], dt_2H));
        if constexpr (WITH_MAX) {
            const double s = __dadd_rn(__dadd_rn(__dmul_rn(sx, sx), __dmul_rn(sy, sy)), __dmul_rn(sz, sz));
            m = (s > m) ? s : m;
        }
    }
    if constexpr (WITH_MAX) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double o = __shfl_xor(m, d, 64);
            m = (o > m) ? o : m;
        }
        if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < WG / 64; ++w) m = (lds[w] > m) ? lds[w] : m;
            block_max[blockIdx.x] = m;
        }
    }
}

// max_i (ax^2 + ay^2 + az^2), per-block partial maxima              src/integrator.cpp:200-213
__global__ void __launch_bounds__(WG) k_max_accel_sq(size_t n, const double* __restrict__ ax,
                                                     const double* __restrict__ ay,
                                                     const double* __restrict__ az,
                                                     double* __restrict__ block_max) {
    __shared__ double lds[WG / 64];
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    double m = 0.0;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double s = __dadd_rn(__dadd_rn(__dmul_rn(ax[i], ax[i]), __dmul_rn(ay[i], ay[i])),
                                   __dmul_rn(az[i], az[i]));
        m = (s > m) ? s : m;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const double o = __shfl_xor(m, d, 64);
        m = (o > m) ? o : m;
    }
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < WG / 64; ++w) m = (lds[w] > m) ? lds[w] : m;
        block_max[blockIdx.x] = m;
    }
}

__global__ void k_fill_int(int n, int* p, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace nbpp
