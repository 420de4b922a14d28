// nbpm_api.hip - libnbpm.so: the mesh (long-range) side of the gravity step on gfx950.
//
// Pipeline of one call (all HBM-bandwidth-bound; meshes are N^3 fp64, z fastest):
//   k_cic_deposit      particles -> rho          8 fp64 atomics per particle (L2 atomics)
//   hipFFT D2Z         rho -> rho_k              N x N x (N/2+1) complex
//   k_greens           phi_k = -4 pi G rho_k/k^2 in place
//   hipFFT Z2D         phi_k -> phi              unnormalised, like the reference's c2r
//   k_gradient         phi -> gx, gy, gz         central differences, periodic
//   k_cic_interpolate  meshes -> acc             8 gathers per mesh per particle
// and, once per run, the initial conditions: white noise -> D2Z -> k_displacement_modes -> 3 x Z2D ->
// displacement meshes -> k_place_particles.
// Each statement follows the reference's operation order so that everything but the two FFTs
// and the order of the deposit's additions is bit for bit the reference's arithmetic
// (compiled with -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>

#include "../../include/nbpm.h"
#include "nb_common.hip.h"

namespace {

static_assert(NBPM_OK == nbc::OK && NBPM_ERR_ARG == nbc::ERR_ARG && NBPM_ERR_HIP == nbc::ERR_HIP &&
              NBPM_ERR_NOINIT == nbc::ERR_NOINIT && NBPM_ERR_NOMEM == nbc::ERR_NOMEM, "error codes");

constexpr int WG = 256;
constexpr double PI = 3.14159265358979323846;   // M_PI

using nbc::Buf;
using nbc::cdiv;
using nbc::fail;

#define FFTCHK(expr)                                                                          \
    do {                                                                                      \
        hipfftResult r_ = (expr);                                                             \
        if (r_ != HIPFFT_SUCCESS)                                                             \
            return fail(NBPM_ERR_HIP, "%s: hipfftResult %d (%s:%d)", #expr, (int)r_, __FILE__, __LINE__); \
    } while (0)

struct Ctx {
    bool ready = false;
    int device = -1;
    hipfftHandle r2c = 0, c2r = 0;
    int plan_n = 0;
    Buf spectrum, phi;                 // rho_k / phi_k (complex), phi (real)
    Buf rho, grav[3];                  // library-owned meshes of nbpm_accelerations_device
    Buf ic_modes[3];                   // displacement coefficients of nbpm_zeldovich_field_device
    Buf host_in[4], host_acc[3];
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool timing = false;
    nbpm_stats stats{};
};
Ctx g;
std::mutex g_mu;

// ---- cloud-in-cell geometry of one particle: src/particles.cpp:60-83 / :112-114 ----
struct Cloud {
    int i0[3], i1[3];
    double f[3];
};
__device__ __forceinline__ Cloud cloud_of(double px, double py, double pz, double cell, int N) {
    Cloud c;
    const double s[3] = {px / cell, py / cell, pz / cell};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int i = static_cast<int>(s[a]);          // :60-62 (truncation)
        c.f[a] = s[a] - i;                             // :64-66
        c.i0[a] = (i + N) % N;                         // :81-83
        c.i1[a] = (i + 1 + N) % N;
    }
    return c;
}
// w000, w100, w010, w110, w001, w101, w011, w111 with the reference's grouping (:69-76)
__device__ __forceinline__ void cloud_weights(const Cloud& c, double (&w)[8]) {
    const double fx = c.f[0], fy = c.f[1], fz = c.f[2];
    w[0] = (1 - fx) * (1 - fy) * (1 - fz);
    w[1] = fx * (1 - fy) * (1 - fz);
    w[2] = (1 - fx) * fy * (1 - fz);
    w[3] = fx * fy * (1 - fz);
    w[4] = (1 - fx) * (1 - fy) * fz;
    w[5] = fx * (1 - fy) * fz;
    w[6] = (1 - fx) * fy * fz;
    w[7] = fx * fy * fz;
}
__device__ __forceinline__ size_t corner(const Cloud& c, int q, int N) {   // q = cx + 2 cy + 4 cz
    const int ix = (q & 1) ? c.i1[0] : c.i0[0];
    const int iy = (q & 2) ? c.i1[1] : c.i0[1];
    const int iz = (q & 4) ? c.i1[2] : c.i0[2];
    return (static_cast<size_t>(ix) * N + iy) * N + iz;
}

// Which WG-sized piece of the particle array this workgroup takes.  Workgroups go to the 8 XCDs
// round-robin and every XCD has its own L2; neighbours in the particle order mostly touch the
// same mesh cells, so within every group of 128 workgroups XCD k takes 16 consecutive pieces
// (deposit at 256^3: 0.77 instead of 0.84 ms).
__device__ __forceinline__ size_t xcd_piece() {
    const size_t b = blockIdx.x, group = b / 128, in_group = b % 128;
    return (group + 1) * 128 <= gridDim.x ? group * 128 + (in_group % 8) * 16 + in_group / 8 : b;
}

// rho += m * w / V.  The reference sums m * w and divides the finished mesh by the cell volume
// (:85-92, :99); here every term is divided, which is the same to rounding and saves a pass.
__global__ void __launch_bounds__(WG) k_cic_deposit(size_t n, const double* __restrict__ x,
                                                    const double* __restrict__ y,
                                                    const double* __restrict__ z,
                                                    const double* __restrict__ m, double cell,
                                                    double cell_volume, int N, double* __restrict__ rho) {
    const size_t i = xcd_piece() * WG + threadIdx.x;
    const bool live = i < n;
    const size_t at = live ? i : n - 1;
    const Cloud c = cloud_of(x[at], y[at], z[at], cell, N);
    double w[8];
    cloud_weights(c, w);
    const double mi = live ? m[at] : 0.0;
    double term[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) term[q] = (mi * w[q]) / cell_volume;
    // Neighbours in the particle order that sit in the same cell add to the same eight mesh cells:
    // once particles have clustered, most of a wavefront's atomics pile onto a few addresses.  Runs
    // of consecutive lanes with the same cell are summed inside the wavefront first (segmented
    // suffix sum) and the first lane of each run issues the atomics.  A wavefront without such a run
    // (the initial lattice) skips all of it.
    const int lane = threadIdx.x & 63;
    const long long key = (static_cast<long long>(c.i0[0]) * N + c.i0[1]) * N + c.i0[2];
    const long long prev = __shfl_up(key, 1, 64);
    const bool head = lane == 0 || prev != key || !live;
    const unsigned long long heads = __ballot(head);
    if (heads != ~0ull) {
        const int run = __popcll(heads & ((2ull << lane) - 1ull));   // 1-based index of this lane's run
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int other_run = __shfl_down(run, d, 64);
            const bool same = lane + d < 64 && other_run == run;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double other = __shfl_down(term[q], d, 64);
                term[q] += same ? other : 0.0;
            }
        }
    }
    // The heads of the runs now hold the runs' sums (every lane is a head where no two neighbours
    // share a cell).  On a lattice-like order the next run mostly sits one cell further along z in
    // the same column of cells, and the upper four corners of one cloud are the lower four of the
    // next: a head adds the upper terms of the run before it to its own lower ones and that run
    // leaves them out, which takes the atomics of a near-uniform set from eight per particle to
    // little over four (the L2's atomic rate is what bounds the kernel: 0.77 -> 0.45 ms at 256^3).
    const unsigned long long below = heads & ((1ull << lane) - 1ull);
    const int prev_head = below ? 63 - __clzll(static_cast<long long>(below)) : 0;
    const long long column = static_cast<long long>(c.i0[0]) * N + c.i0[1];
    const long long prev_column = __shfl(column, prev_head, 64);
    const int prev_upper = __shfl(c.i1[2], prev_head, 64);
    const int prev_live = __shfl(live ? 1 : 0, prev_head, 64);
    const bool link = head && lane > 0 && live && prev_live && prev_column == column && prev_upper == c.i0[2];
    const unsigned long long links = __ballot(link);
    if (links) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double up = __shfl(term[q + 4], prev_head, 64);
            term[q] += link ? up : 0.0;
        }
    }
    if (!live || !head) return;
    const unsigned long long above = lane < 63 ? heads >> (lane + 1) : 0ull;    // the next run's head, if any
    const bool taken = above && ((links >> (lane + 1 + __builtin_ctzll(above))) & 1ull);
#pragma unroll
    for (int q = 0; q < 4; ++q) unsafeAtomicAdd(&rho[corner(c, q, N)], term[q]);
    if (taken) return;
#pragma unroll
    for (int q = 4; q < 8; ++q) unsafeAtomicAdd(&rho[corner(c, q, N)], term[q]);
}

// phi_k = rho_k * (-4 pi G / k^2), phi_0 = 0                      src/integrator.cpp:78-103
__global__ void __launch_bounds__(WG) k_greens(int N, double L, double G, double2* __restrict__ spec) {
    const int nk = N / 2 + 1;
    const size_t total = static_cast<size_t>(N) * N * nk;
    const size_t idx = static_cast<size_t>(blockIdx.x) * WG + threadIdx.x;
    if (idx >= total) return;
    const int k = static_cast<int>(idx % nk);
    const int j = static_cast<int>((idx / nk) % N);
    const int i = static_cast<int>(idx / (static_cast<size_t>(nk) * N));
    if (i == 0 && j == 0 && k == 0) {
        spec[0] = make_double2(0.0, 0.0);
        return;
    }
    const double kx_freq = static_cast<double>((i < N / 2) ? i : (i - N));
    const double ky_freq = static_cast<double>((j < N / 2) ? j : (j - N));
    const double kz_freq = static_cast<double>(k);
    const double kx = kx_freq * 2.0 * PI / L;
    const double ky = ky_freq * 2.0 * PI / L;
    const double kz = kz_freq * 2.0 * PI / L;
    const double k2 = kx * kx + ky * ky + kz * kz;
    const double f = -4.0 * PI * G / k2;
    const double2 v = spec[idx];
    spec[idx] = make_double2(v.x * f, v.y * f);
}

// g = (phi(+1) - phi(-1)) * factor along each axis, periodic            src/integrator.cpp:108-131
__global__ void __launch_bounds__(WG) k_gradient(int N, double factor, const double* __restrict__ phi,
                                                 double* __restrict__ gx, double* __restrict__ gy,
                                                 double* __restrict__ gz) {
    const size_t total = static_cast<size_t>(N) * N * N;
    const size_t idx = static_cast<size_t>(blockIdx.x) * WG + threadIdx.x;
    if (idx >= total) return;
    const int k = static_cast<int>(idx % N);
    const int j = static_cast<int>((idx / N) % N);
    const int i = static_cast<int>(idx / (static_cast<size_t>(N) * N));
    const int ip1 = (i + 1 == N) ? 0 : i + 1, im1 = (i == 0) ? N - 1 : i - 1;
    const int jp1 = (j + 1 == N) ? 0 : j + 1, jm1 = (j == 0) ? N - 1 : j - 1;
    const int kp1 = (k + 1 == N) ? 0 : k + 1, km1 = (k == 0) ? N - 1 : k - 1;
    auto at = [&](int a, int b, int c) { return phi[(static_cast<size_t>(a) * N + b) * N + c]; };
    gx[idx] = (at(ip1, j, k) - at(im1, j, k)) * factor;
    gy[idx] = (at(i, jp1, k) - at(i, jm1, k)) * factor;
    gz[idx] = (at(i, j, kp1) - at(i, j, km1)) * factor;
}

// acc = sum of the 8 corner values * weights, in the reference's order   src/particles.cpp:116-129
__global__ void __launch_bounds__(WG) k_cic_interpolate(size_t n, const double* __restrict__ x,
                                                        const double* __restrict__ y,
                                                        const double* __restrict__ z, double cell, int N,
                                                        const double* __restrict__ gx,
                                                        const double* __restrict__ gy,
                                                        const double* __restrict__ gz,
                                                        double* __restrict__ ax, double* __restrict__ ay,
                                                        double* __restrict__ az) {
    const size_t i = static_cast<size_t>(blockIdx.x) * WG + threadIdx.x;   // (xcd_piece() measured: no gain here)
    if (i >= n) return;
    const Cloud c = cloud_of(x[i], y[i], z[i], cell, N);
    double w[8];
    cloud_weights(c, w);
    size_t at[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) at[q] = corner(c, q, N);
    auto interp = [&](const double* __restrict__ grid) {
        double s = grid[at[0]] * w[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) s = s + grid[at[q]] * w[q];
        return s;
    };
    ax[i] = interp(gx);
    ay[i] = interp(gy);
    az[i] = interp(gz);
}

// ---- initial conditions: Zel'dovich displacement field and the particles on it -------------------
// Square root of the matter power spectrum's SHAPE at wavenumber kh (h / Mpc): kh^(n/2) T(q) with
// the BBKS transfer function T(q) = ln(1 + 2.34 q) / (2.34 q) [1 + 3.89 q + (16.1 q)^2 + (5.46 q)^3
// + (6.71 q)^4]^(-1/4), q = kh / (Omega_m h)   (the spectrum of src/ics.cpp:11-27, whose P(k) is the
// square of this).  One function for the host quadrature and the mode kernel.
struct SpectrumShape {
    double inv_gamma;      // 1 / (Omega_m h)
    double half_index;     // n_s / 2
};
__host__ __device__ inline double sqrt_power_shape(const SpectrumShape& sp, double kh) {
    const double q = kh * sp.inv_gamma;
    const double t = 2.34 * q;
    // the bracket as a polynomial in q, by Horner's rule
    constexpr double c2 = 16.1 * 16.1, c3 = 5.46 * 5.46 * 5.46, c4 = (6.71 * 6.71) * (6.71 * 6.71);
    const double bracket = 1.0 + q * (3.89 + q * (c2 + q * (c3 + q * c4)));
    const double transfer = log1p(t) / (t * sqrt(sqrt(bracket)));
    return pow(kh, sp.half_index) * transfer;
}

// Fourier transform of the spherical top hat of radius R at x = k R.
inline double tophat_window(double x) {
    if (x < 1e-4) return 1.0 - x * x / 10.0;                // the series, where 3 (sin x - x cos x) / x^3 cancels
    return 3.0 * (std::sin(x) - x * std::cos(x)) / (x * x * x);
}

// One thread per Fourier mode (i, j, k) of the half spectrum: the white-noise coefficient W becomes
// the three displacement coefficients  D_c = i c_c / |c|^2 * sqrt(P(|k|)) * W * scale,  with c the
// wave vector in code units (2 pi f / L), |k| the same vector in h / Mpc, and `scale` everything that
// does not depend on the mode (sqrt(A M^3 / V_box) for the amplitude of a discrete field, a / M^3
// for the growth factor and the unnormalised inverse transform).  i W = (-Im W, Re W).  The mean
// mode is zero.  Frequencies follow src/ics.cpp:95-97: index M/2 counts as -M/2 along x and y and as
// +M/2 along z (the half spectrum's last plane).
__global__ void __launch_bounds__(WG) k_displacement_modes(int M, SpectrumShape sp, double kh_per_freq,
                                                           double c_per_freq, double scale,
                                                           const double2* __restrict__ noise,
                                                           double2* __restrict__ dxk, double2* __restrict__ dyk,
                                                           double2* __restrict__ dzk) {
    const int nk = M / 2 + 1;
    const size_t total = static_cast<size_t>(M) * M * nk;
    const size_t idx = static_cast<size_t>(blockIdx.x) * WG + threadIdx.x;
    if (idx >= total) return;
    const int k = static_cast<int>(idx % nk);
    const int j = static_cast<int>((idx / nk) % M);
    const int i = static_cast<int>(idx / (static_cast<size_t>(nk) * M));
    const double f[3] = {static_cast<double>(i < M / 2 ? i : i - M), static_cast<double>(j < M / 2 ? j : j - M),
                         static_cast<double>(k)};
    const double f2 = f[0] * f[0] + f[1] * f[1] + f[2] * f[2];      // exact: small integers
    double g = 0.0;
    if (f2 > 0.0) {
        const double kh = sqrt(f2) * kh_per_freq;
        g = scale * sqrt_power_shape(sp, kh) / (f2 * c_per_freq * c_per_freq);
    }
    const double2 w = noise[idx];
    const double re = -w.y * g, im = w.x * g;
    dxk[idx] = make_double2(f[0] * c_per_freq * re, f[0] * c_per_freq * im);
    dyk[idx] = make_double2(f[1] * c_per_freq * re, f[1] * c_per_freq * im);
    dzk[idx] = make_double2(f[2] * c_per_freq * re, f[2] * c_per_freq * im);
}

// One thread per particle p = (a N + b) N + c of the N^3 lattice: it starts at the centre of its
// lattice cell, is displaced by the field value of the MESH cell that centre lies in (truncation,
// no interpolation: src/ics.cpp:181-191), folded into [0, L) with fmod(q + d + L, L) (:193-198) and
// given the velocity `velocity_factor` * d (H f d, or 0 for particles at rest: :200-208).
__global__ void __launch_bounds__(WG) k_place_particles(int N, int M, double L, double velocity_factor, double mass,
                                                        const double* __restrict__ dx, const double* __restrict__ dy,
                                                        const double* __restrict__ dz, double* __restrict__ x,
                                                        double* __restrict__ y, double* __restrict__ z,
                                                        double* __restrict__ vx, double* __restrict__ vy,
                                                        double* __restrict__ vz, double* __restrict__ m) {
    const size_t total = static_cast<size_t>(N) * N * N;
    const size_t p = static_cast<size_t>(blockIdx.x) * WG + threadIdx.x;
    if (p >= total) return;
    const int lattice[3] = {static_cast<int>(p / (static_cast<size_t>(N) * N)), static_cast<int>((p / N) % N),
                            static_cast<int>(p % N)};
    const double spacing = L / N, cell = L / M;
    double q[3];
    size_t at = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        q[a] = (lattice[a] + 0.5) * spacing;
        at = at * M + static_cast<size_t>(static_cast<int>(q[a] / cell) % M);
    }
    const double d[3] = {dx[at], dy[at], dz[at]};
    x[p] = fmod(q[0] + d[0] + L, L);
    y[p] = fmod(q[1] + d[1] + L, L);
    z[p] = fmod(q[2] + d[2] + L, L);
    vx[p] = velocity_factor * d[0];
    vy[p] = velocity_factor * d[1];
    vz[p] = velocity_factor * d[2];
    if (m) m[p] = mass;
}

int check_ic_params(const nbpm_ic_params* p) {
    if (!p) return fail(NBPM_ERR_ARG, "params is NULL");
    if (p->mesh_size < 2 || p->mesh_size > 1290) return fail(NBPM_ERR_ARG, "mesh_size must be in [2, 1290] (got %d)", p->mesh_size);
    if (p->n_per_side < 0 || p->n_per_side > 1290) return fail(NBPM_ERR_ARG, "n_per_side must be in [0, 1290] (got %d)", p->n_per_side);
    for (double v : {p->domain_size, p->box_size_mpc, p->hubble_param, p->omega_m})
        if (!(v > 0.0) || !std::isfinite(v)) return fail(NBPM_ERR_ARG, "domain_size, box_size_mpc, hubble_param and omega_m must be positive and finite");
    if (!std::isfinite(p->spectral_index) || !(p->amplitude >= 0.0) || !std::isfinite(p->amplitude) || !std::isfinite(p->scale_factor))
        return fail(NBPM_ERR_ARG, "spectral_index, amplitude (>= 0) and scale_factor must be finite");
    return NBPM_OK;
}

int check_params(const nbpm_params* p) {
    if (!p) return fail(NBPM_ERR_ARG, "params is NULL");
    if (!(p->domain_size > 0.0) || !std::isfinite(p->domain_size))
        return fail(NBPM_ERR_ARG, "domain_size must be positive and finite");
    if (!std::isfinite(p->G)) return fail(NBPM_ERR_ARG, "G must be finite");
    if (p->mesh_size < 2 || p->mesh_size > 1290)
        return fail(NBPM_ERR_ARG, "mesh_size must be in [2, 1290] (got %d)", p->mesh_size);
    return NBPM_OK;
}

int ensure_plans(int N, hipStream_t s) {
    if (g.plan_n != N) {
        if (g.plan_n) {
            (void)hipfftDestroy(g.r2c);
            (void)hipfftDestroy(g.c2r);
            g.plan_n = 0;
        }
        FFTCHK(hipfftPlan3d(&g.r2c, N, N, N, HIPFFT_D2Z));
        if (hipfftPlan3d(&g.c2r, N, N, N, HIPFFT_Z2D) != HIPFFT_SUCCESS) {
            (void)hipfftDestroy(g.r2c);
            return fail(NBPM_ERR_HIP, "hipfftPlan3d(Z2D, %d^3) failed", N);
        }
        g.plan_n = N;
    }
    FFTCHK(hipfftSetStream(g.r2c, s));
    FFTCHK(hipfftSetStream(g.c2r, s));
    return NBPM_OK;
}

int deposit(const nbpm_params* p, size_t n, const double* x, const double* y, const double* z,
            const double* m, double* rho, hipStream_t s) {
    const int N = p->mesh_size;
    const size_t cells = static_cast<size_t>(N) * N * N;
    const double cell = p->domain_size / N;              // CELL_SIZE   (src/config.cpp:74)
    const double volume = cell * cell * cell;            // CELL_VOLUME (src/config.cpp:75)
    HIPCHK(hipMemsetAsync(rho, 0, sizeof(double) * cells, s));
    if (n) {
        hipLaunchKernelGGL(k_cic_deposit, dim3(cdiv(n, WG)), dim3(WG), 0, s, n, x, y, z, m, cell, volume, N, rho);
        HIPCHK(hipGetLastError());
    }
    return NBPM_OK;
}

int solve(const nbpm_params* p, const double* rho, double* gx, double* gy, double* gz, hipStream_t s) {
    const int N = p->mesh_size;
    const size_t cells = static_cast<size_t>(N) * N * N;
    const size_t modes = static_cast<size_t>(N) * N * (N / 2 + 1);
    CHK(ensure_plans(N, s));
    CHK(g.spectrum.ensure(sizeof(double2) * modes));
    CHK(g.phi.ensure(sizeof(double) * cells));
    double2* spec = g.spectrum.as<double2>();
    double* phi = g.phi.as<double>();
    // out-of-place, so the caller's density survives
    FFTCHK(hipfftExecD2Z(g.r2c, const_cast<double*>(rho), reinterpret_cast<hipfftDoubleComplex*>(spec)));
    hipLaunchKernelGGL(k_greens, dim3(cdiv(modes, WG)), dim3(WG), 0, s, N, p->domain_size, p->G, spec);
    HIPCHK(hipGetLastError());
    FFTCHK(hipfftExecZ2D(g.c2r, reinterpret_cast<hipfftDoubleComplex*>(spec), phi));
    const double cell = p->domain_size / N;
    const double norm = 1.0 / (static_cast<double>(N) * N * N);   // src/integrator.cpp:108
    const double factor = -norm / (2.0 * cell);                   // :109
    hipLaunchKernelGGL(k_gradient, dim3(cdiv(cells, WG)), dim3(WG), 0, s, N, factor, phi, gx, gy, gz);
    HIPCHK(hipGetLastError());
    return NBPM_OK;
}

int interpolate(const nbpm_params* p, size_t n, const double* x, const double* y, const double* z,
                const double* gx, const double* gy, const double* gz, double* ax, double* ay,
                double* az, hipStream_t s) {
    if (!n) return NBPM_OK;
    const int N = p->mesh_size;
    const double cell = p->domain_size / N;
    hipLaunchKernelGGL(k_cic_interpolate, dim3(cdiv(n, WG)), dim3(WG), 0, s, n, x, y, z, cell, N, gx, gy, gz,
                       ax, ay, az);
    HIPCHK(hipGetLastError());
    return NBPM_OK;
}

int accelerations(const nbpm_params* p, size_t n, const double* x, const double* y, const double* z,
                  const double* m, double* ax, double* ay, double* az, hipStream_t s) {
    const size_t cells = static_cast<size_t>(p->mesh_size) * p->mesh_size * p->mesh_size;
    CHK(g.rho.ensure(sizeof(double) * cells));
    for (Buf& b : g.grav) CHK(b.ensure(sizeof(double) * cells));
    if (g.timing) HIPCHK(hipEventRecord(g.ev[0], s));
    CHK(deposit(p, n, x, y, z, m, g.rho.as<double>(), s));
    if (g.timing) HIPCHK(hipEventRecord(g.ev[1], s));
    CHK(solve(p, g.rho.as<double>(), g.grav[0].as<double>(), g.grav[1].as<double>(), g.grav[2].as<double>(), s));
    if (g.timing) HIPCHK(hipEventRecord(g.ev[2], s));
    CHK(interpolate(p, n, x, y, z, g.grav[0].as<double>(), g.grav[1].as<double>(), g.grav[2].as<double>(),
                    ax, ay, az, s));
    if (g.timing) {
        HIPCHK(hipEventRecord(g.ev[3], s));
        HIPCHK(hipEventSynchronize(g.ev[3]));
        float a = 0, b = 0, c = 0;
        HIPCHK(hipEventElapsedTime(&a, g.ev[0], g.ev[1]));
        HIPCHK(hipEventElapsedTime(&b, g.ev[1], g.ev[2]));
        HIPCHK(hipEventElapsedTime(&c, g.ev[2], g.ev[3]));
        g.stats.deposit_ms = a;
        g.stats.solve_ms = b;
        g.stats.interpolate_ms = c;
        g.stats.total_ms = a + b + c;
    }
    return NBPM_OK;
}

#define ENTER()                                                              \
    std::lock_guard<std::mutex> lock_(g_mu);                                 \
    if (!g.ready) return fail(NBPM_ERR_NOINIT, "nbpm_init has not been called"); \
    HIPCHK(hipSetDevice(g.device))

}  // namespace

extern "C" {

int nbpm_init(int device) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g.ready) {
        if (g.device == device) return NBPM_OK;
        return fail(NBPM_ERR_ARG, "already initialised on device %d", g.device);
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        (void)hipGetLastError();
        return fail(NBPM_ERR_HIP, "no HIP device is visible; libnbpm has no CPU path");
    }
    if (device < 0 || device >= count) return fail(NBPM_ERR_ARG, "device %d out of range [0, %d)", device, count);
    HIPCHK(hipSetDevice(device));
    for (hipEvent_t& e : g.ev) HIPCHK(hipEventCreate(&e));
    g.device = device;
    g.ready = true;
    return NBPM_OK;
}

void nbpm_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g.ready) return;
    (void)hipSetDevice(g.device);
    (void)hipDeviceSynchronize();
    if (g.plan_n) {
        (void)hipfftDestroy(g.r2c);
        (void)hipfftDestroy(g.c2r);
        g.plan_n = 0;
    }
    for (Buf* b : {&g.spectrum, &g.phi, &g.rho, &g.grav[0], &g.grav[1], &g.grav[2], &g.host_in[0],
                   &g.host_in[1], &g.host_in[2], &g.host_in[3], &g.host_acc[0], &g.host_acc[1], &g.host_acc[2],
                   &g.ic_modes[0], &g.ic_modes[1], &g.ic_modes[2]})
        b->release();
    for (hipEvent_t& e : g.ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    g.ready = false;
    g.device = -1;
}

const char* nbpm_last_error(void) { return nbc::error_text(); }

int nbpm_set_timing(int on) {
    std::lock_guard<std::mutex> lock(g_mu);
    g.timing = on != 0;
    return NBPM_OK;
}

int nbpm_get_stats(nbpm_stats* out) {
    if (!out) return fail(NBPM_ERR_ARG, "out is NULL");
    std::lock_guard<std::mutex> lock(g_mu);
    *out = g.stats;
    return NBPM_OK;
}

int nbpm_deposit_device(const nbpm_params* p, size_t n, const double* x, const double* y,
                        const double* z, const double* m, double* rho, void* stream) {
    ENTER();
    CHK(check_params(p));
    if (!rho || (n && (!x || !y || !z || !m))) return fail(NBPM_ERR_ARG, "NULL array");
    return deposit(p, n, x, y, z, m, rho, static_cast<hipStream_t>(stream));
}

int nbpm_solve_device(const nbpm_params* p, const double* rho, double* gx, double* gy,
                      double* gz, void* stream) {
    ENTER();
    CHK(check_params(p));
    if (!rho || !gx || !gy || !gz) return fail(NBPM_ERR_ARG, "NULL array");
    return solve(p, rho, gx, gy, gz, static_cast<hipStream_t>(stream));
}

int nbpm_potential_device(const nbpm_params* p, double* phi, void* stream) {
    ENTER();
    CHK(check_params(p));
    if (!phi) return fail(NBPM_ERR_ARG, "phi is NULL");
    const size_t cells = static_cast<size_t>(p->mesh_size) * p->mesh_size * p->mesh_size;
    if (g.plan_n != p->mesh_size || g.phi.cap < sizeof(double) * cells)
        return fail(NBPM_ERR_ARG, "no potential of a %d^3 mesh has been computed", p->mesh_size);
    HIPCHK(hipMemcpyAsync(phi, g.phi.p, sizeof(double) * cells, hipMemcpyDeviceToDevice,
                          static_cast<hipStream_t>(stream)));
    return NBPM_OK;
}

int nbpm_interpolate_device(const nbpm_params* p, size_t n, const double* x, const double* y,
                            const double* z, const double* gx, const double* gy,
                            const double* gz, double* ax, double* ay, double* az,
                            void* stream) {
    ENTER();
    CHK(check_params(p));
    if (!gx || !gy || !gz || (n && (!x || !y || !z || !ax || !ay || !az)))
        return fail(NBPM_ERR_ARG, "NULL array");
    return interpolate(p, n, x, y, z, gx, gy, gz, ax, ay, az, static_cast<hipStream_t>(stream));
}

int nbpm_accelerations_device(const nbpm_params* p, size_t n, const double* x, const double* y,
                              const double* z, const double* m, double* ax, double* ay,
                              double* az, void* stream) {
    ENTER();
    CHK(check_params(p));
    if (n && (!x || !y || !z || !m || !ax || !ay || !az)) return fail(NBPM_ERR_ARG, "NULL array");
    return accelerations(p, n, x, y, z, m, ax, ay, az, static_cast<hipStream_t>(stream));
}

int nbpm_sigma8_amplitude(double sigma_8, double omega_m, double hubble_param, double spectral_index, double* amplitude) {
    if (!amplitude) return fail(NBPM_ERR_ARG, "amplitude is NULL");
    if (!(omega_m > 0.0) || !(hubble_param > 0.0) || !std::isfinite(sigma_8) || !std::isfinite(spectral_index))
        return fail(NBPM_ERR_ARG, "omega_m and hubble_param must be positive, sigma_8 and spectral_index finite");
    // variance of the unnormalised spectrum in spheres of 8 Mpc/h: (1 / 2 pi^2) * integral of P W^2 k^2 dk
    // by the rectangle rule the reference defines its normalisation with (src/ics.cpp:38-51): nodes
    // k = 0.001, 0.002, ... < 100, weight 0.001 each.  The terms are summed pairwise (blocks of 64).
    const SpectrumShape sp{1.0 / (omega_m * hubble_param), 0.5 * spectral_index};
    constexpr double dk = 0.001, R = 8.0;
    constexpr int NODES = 99999;                            // 0.001 * 99999 = 99.999 < 100
    double total = 0.0, block = 0.0;
    for (int i = 1; i <= NODES; ++i) {
        const double k = i * dk;
        const double root = sqrt_power_shape(sp, k), w = tophat_window(k * R);
        block += root * root * w * w * k * k * dk;
        if (i % 64 == 0 || i == NODES) {
            total += block;
            block = 0.0;
        }
    }
    *amplitude = sigma_8 * sigma_8 / (total / (2.0 * PI * PI));
    return NBPM_OK;
}

int nbpm_zeldovich_field_device(const nbpm_ic_params* p, const double* white_noise, double* dx, double* dy,
                                double* dz, void* stream) {
    ENTER();
    CHK(check_ic_params(p));
    if (!white_noise || !dx || !dy || !dz) return fail(NBPM_ERR_ARG, "NULL array");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int M = p->mesh_size;
    const size_t modes = static_cast<size_t>(M) * M * (M / 2 + 1);
    CHK(ensure_plans(M, s));
    CHK(g.spectrum.ensure(sizeof(double2) * modes));
    for (Buf& b : g.ic_modes) CHK(b.ensure(sizeof(double2) * modes));
    double2* noise_k = g.spectrum.as<double2>();
    FFTCHK(hipfftExecD2Z(g.r2c, const_cast<double*>(white_noise), reinterpret_cast<hipfftDoubleComplex*>(noise_k)));
    const SpectrumShape sp{1.0 / (p->omega_m * p->hubble_param), 0.5 * p->spectral_index};
    const double cells = static_cast<double>(M) * M * M;
    const double box_volume = p->box_size_mpc * p->box_size_mpc * p->box_size_mpc;
    // per-mode constants: amplitude of a discrete Gaussian field with spectrum A P(k) on M^3 cells of
    // a box of that volume; growth factor a; 1 / M^3 of the unnormalised inverse transform
    const double scale = std::sqrt(p->amplitude * cells / box_volume) * (p->scale_factor / cells);
    hipLaunchKernelGGL(k_displacement_modes, dim3(cdiv(modes, WG)), dim3(WG), 0, s, M, sp,
                       2.0 * PI / (p->box_size_mpc * p->hubble_param), 2.0 * PI / p->domain_size, scale, noise_k,
                       g.ic_modes[0].as<double2>(), g.ic_modes[1].as<double2>(), g.ic_modes[2].as<double2>());
    HIPCHK(hipGetLastError());
    double* out[3] = {dx, dy, dz};
    for (int c = 0; c < 3; ++c)
        FFTCHK(hipfftExecZ2D(g.c2r, reinterpret_cast<hipfftDoubleComplex*>(g.ic_modes[c].as<double2>()), out[c]));
    return NBPM_OK;
}

int nbpm_zeldovich_particles_device(const nbpm_ic_params* p, const double* dx, const double* dy, const double* dz,
                                    double velocity_factor, double mass, double* x, double* y, double* z, double* vx,
                                    double* vy, double* vz, double* m, void* stream) {
    ENTER();
    CHK(check_ic_params(p));
    const size_t n = static_cast<size_t>(p->n_per_side) * p->n_per_side * p->n_per_side;
    if (n == 0) return NBPM_OK;
    if (!dx || !dy || !dz || !x || !y || !z || !vx || !vy || !vz) return fail(NBPM_ERR_ARG, "NULL array");
    hipLaunchKernelGGL(k_place_particles, dim3(cdiv(n, WG)), dim3(WG), 0, static_cast<hipStream_t>(stream), p->n_per_side,
                       p->mesh_size, p->domain_size, velocity_factor, mass, dx, dy, dz, x, y, z, vx, vy, vz, m);
    HIPCHK(hipGetLastError());
    return NBPM_OK;
}

int nbpm_accelerations(const nbpm_params* p, size_t n, const double* x, const double* y,
                       const double* z, const double* m, double* ax, double* ay, double* az) {
    ENTER();
    CHK(check_params(p));
    if (n == 0) return NBPM_OK;
    if (!x || !y || !z || !m || !ax || !ay || !az) return fail(NBPM_ERR_ARG, "NULL array");
    const size_t bytes = sizeof(double) * n;
    const double* in[4] = {x, y, z, m};
    for (int c = 0; c < 4; ++c) {
        CHK(g.host_in[c].ensure(bytes));
        HIPCHK(hipMemcpyAsync(g.host_in[c].p, in[c], bytes, hipMemcpyHostToDevice, nullptr));
    }
    for (Buf& b : g.host_acc) CHK(b.ensure(bytes));
    CHK(accelerations(p, n, g.host_in[0].as<double>(), g.host_in[1].as<double>(), g.host_in[2].as<double>(),
                      g.host_in[3].as<double>(), g.host_acc[0].as<double>(), g.host_acc[1].as<double>(),
                      g.host_acc[2].as<double>(), nullptr));
    double* out[3] = {ax, ay, az};
    for (int c = 0; c < 3; ++c) HIPCHK(hipMemcpyAsync(out[c], g.host_acc[c].p, bytes, hipMemcpyDeviceToHost, nullptr));
    HIPCHK(hipStreamSynchronize(nullptr));
    return NBPM_OK;
}

}  // extern "C"
