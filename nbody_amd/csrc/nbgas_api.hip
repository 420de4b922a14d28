// nbgas_api.hip - libnbgas.so: the gas side of the step on gfx950.
//
// The reference's solver (src/gas.cpp) is a chain of whole-mesh array statements: some sixty
// temporaries of N^3 doubles per sweep.  Everything in it is a pure function of a cell's five
// conserved values and those of its two neighbours along the sweep axis, so a sweep is ONE
// kernel here: a thread recomputes the primitive variables of its cell and of both neighbours,
// forms the two face fluxes and writes the five updated conserved values (40 B read + 40 B
// written per cell, HBM-bound; the neighbours come from cache).  Sweeps ping-pong between the
// caller's meshes and one library-owned set, and the last kernel brings all nine meshes up to
// date.  Each expression keeps the reference's operation order (-ffp-contract=off).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <limits>
#include <mutex>

#include "../../include/nbgas.h"
#include "nb_common.hip.h"

namespace {

static_assert(NBGAS_OK == nbc::OK && NBGAS_ERR_ARG == nbc::ERR_ARG && NBGAS_ERR_HIP == nbc::ERR_HIP &&
              NBGAS_ERR_NOINIT == nbc::ERR_NOINIT && NBGAS_ERR_NOMEM == nbc::ERR_NOMEM, "error codes");

constexpr int WG = 256;
constexpr double FLOOR = 1e-12;
constexpr unsigned PINNED_DOUBLES = 16384;   // partial maxima the CFL read-back can carry (128 KiB)

using nbc::Buf;
using nbc::cdiv;
using nbc::fail;

struct Ctx {
    bool ready = false;
    int device = -1;
    Buf scratch;     // five conserved meshes: the other side of the sweeps' ping-pong
    Buf blockmax;
    void* pinned = nullptr;   // PINNED_DOUBLES of page-locked host memory for the CFL read-back
    hipEvent_t cfl_done = nullptr;   // the read-back of nbgas_cfl_timestep_enqueue_device has landed
    unsigned cfl_blocks = 0;         // partial maxima in flight (0: nothing enqueued)
};
Ctx g;
std::mutex g_mu;

// ---- one cell --------------------------------------------------------------------------
struct Conserved { double rho, m[3], en; };
struct Cell {                    // conserved values with the energy re-synchronised + primitives
    double rho, m[3], en, v[3], p;
};

// GasGrid::update_primitive_variables for one cell (src/gas.cpp:210-246)
__device__ __forceinline__ Cell primitives(const Conserved& u, double gm1, double inv_gm1) {
    Cell c;
    c.rho = u.rho;
    c.m[0] = u.m[0]; c.m[1] = u.m[1]; c.m[2] = u.m[2];
    if (u.rho > FLOOR) {
        const double inv_rho = 1.0 / u.rho;
        c.v[0] = u.m[0] * inv_rho;
        c.v[1] = u.m[1] * inv_rho;
        c.v[2] = u.m[2] * inv_rho;
        const double ke = 0.5 * u.rho * (c.v[0] * c.v[0] + c.v[1] * c.v[1] + c.v[2] * c.v[2]);
        double p = gm1 * (u.en - ke);
        if (p < FLOOR) p = FLOOR;
        c.p = p;
        c.en = p * inv_gm1 + ke;
    } else {
        c.v[0] = c.v[1] = c.v[2] = 0.0;
        c.p = FLOOR;
        c.en = FLOOR * inv_gm1;
    }
    return c;
}

// RiemannSolver::solve_hll for one face (src/gas.cpp:53-64)
__device__ __forceinline__ double hll(double FL, double FR, double UL, double UR, double SL, double SR,
                                      double SR_minus_SL) {
    if (SL >= 0) return FL;
    if (SR <= 0) return FR;
    return (SR * FL - SL * FR + SL * SR * (UR - UL)) / SR_minus_SL;
}

// The five fluxes through the face between L and its +axis neighbour R, in the sweep's frame:
// n = normal component, t1 / t2 = transverse (src/gas.cpp:125-166).  out: density, mom_n,
// mom_t1, mom_t2, energy.
__device__ __forceinline__ void face_flux(const Cell& L, const Cell& R, int n, int t1, int t2, double gamma,
                                          double (&out)[5]) {
    double cs_L = sqrt(gamma * L.p / L.rho), cs_R = sqrt(gamma * R.p / R.rho);
    cs_L = (L.rho > FLOOR) ? cs_L : 0.0;
    cs_R = (R.rho > FLOOR) ? cs_R : 0.0;
    cs_L = (cs_L == cs_L) ? cs_L : 0.0;
    cs_R = (cs_R == cs_R) ? cs_R : 0.0;
    const double a = L.v[n] - cs_L, b = R.v[n] - cs_R;
    const double SL = (b < a) ? b : a;                    // cwiseMin
    const double c = L.v[n] + cs_L, d = R.v[n] + cs_R;
    const double SR = (c < d) ? d : c;                    // cwiseMax
    double dS = SR - SL;
    dS = (fabs(dS) < 1e-9) ? 1e-9 : dS;
    out[0] = hll(L.rho * L.v[n], R.rho * R.v[n], L.rho, R.rho, SL, SR, dS);
    out[1] = hll(L.rho * (L.v[n] * L.v[n]) + L.p, R.rho * (R.v[n] * R.v[n]) + R.p, L.m[n], R.m[n], SL, SR, dS);
    out[2] = hll(L.rho * L.v[n] * L.v[t1], R.rho * R.v[n] * R.v[t1], L.m[t1], R.m[t1], SL, SR, dS);
    out[3] = hll(L.rho * L.v[n] * L.v[t2], R.rho * R.v[n] * R.v[t2], L.m[t2], R.m[t2], SL, SR, dS);
    out[4] = hll((L.en + L.p) * L.v[n], (R.en + R.p) * R.v[n], L.en, R.en, SL, SR, dS);
}

struct Fields {   // five conserved meshes
    double* f[5];   // density, momentum_x, momentum_y, momentum_z, energy
};

__device__ __forceinline__ Conserved load_cell(const Fields& u, size_t i) {
    Conserved c;
    c.rho = u.f[0][i]; c.m[0] = u.f[1][i]; c.m[1] = u.f[2][i]; c.m[2] = u.f[3][i]; c.en = u.f[4][i];
    return c;
}

// One sweep along AXIS (src/gas.cpp:253-294): out = in' - factor (F(+face) - F(-face)), in' =
// in with the energy re-synchronised by update_primitive_variables.
template <int AXIS>
__global__ void __launch_bounds__(WG) k_sweep(int N, double factor, double gamma, Fields in, Fields out) {
    const size_t total = static_cast<size_t>(N) * N * N;
    const size_t idx = static_cast<size_t>(blockIdx.x) * WG + threadIdx.x;
    if (idx >= total) return;
    const int k = static_cast<int>(idx % N);
    const int j = static_cast<int>((idx / N) % N);
    const int i = static_cast<int>(idx / (static_cast<size_t>(N) * N));
    const int pos = AXIS == 0 ? i : (AXIS == 1 ? j : k);
    const size_t stride = AXIS == 0 ? static_cast<size_t>(N) * N : (AXIS == 1 ? N : 1);
    const size_t up = (pos + 1 == N) ? idx - stride * (N - 1) : idx + stride;
    const size_t dn = (pos == 0) ? idx + stride * (N - 1) : idx - stride;
    constexpr int n = AXIS, t1 = (AXIS + 1) % 3, t2 = (AXIS + 2) % 3;   // src/gas.cpp:74-98
    const double gm1 = gamma - 1.0, inv_gm1 = 1.0 / gm1;
    const Cell me = primitives(load_cell(in, idx), gm1, inv_gm1);
    const Cell hi = primitives(load_cell(in, up), gm1, inv_gm1);
    const Cell lo = primitives(load_cell(in, dn), gm1, inv_gm1);
    double fp[5], fm[5];
    face_flux(me, hi, n, t1, t2, gamma, fp);    // flux(idx)
    face_flux(lo, me, n, t1, t2, gamma, fm);    // flux rolled by +1: the face below
    double rho = me.rho - factor * (fp[0] - fm[0]);
    rho = (rho < FLOOR) ? FLOOR : rho;
    out.f[0][idx] = rho;
    out.f[1 + n][idx] = me.m[n] - factor * (fp[1] - fm[1]);
    out.f[1 + t1][idx] = me.m[t1] - factor * (fp[2] - fm[2]);
    out.f[1 + t2][idx] = me.m[t2] - factor * (fp[3] - fm[3]);
    out.f[4][idx] = me.en - factor * (fp[4] - fm[4]);
}

// update_primitive_variables over a mesh: reads `in`, writes all nine meshes of `s`
__global__ void __launch_bounds__(WG) k_primitives(size_t total, double gamma, Fields in, nbgas_state s) {
    const size_t idx = static_cast<size_t>(blockIdx.x) * WG + threadIdx.x;
    if (idx >= total) return;
    const double gm1 = gamma - 1.0, inv_gm1 = 1.0 / gm1;
    const Cell c = primitives(load_cell(in, idx), gm1, inv_gm1);
    s.density[idx] = c.rho;
    s.momentum_x[idx] = c.m[0]; s.momentum_y[idx] = c.m[1]; s.momentum_z[idx] = c.m[2];
    s.energy[idx] = c.en;
    s.pressure[idx] = c.p;
    s.velocity_x[idx] = c.v[0]; s.velocity_y[idx] = c.v[1]; s.velocity_z[idx] = c.v[2];
}

// per-block maxima of |v| + c_s from the stored primitives (src/gas.cpp:304-318)
__global__ void __launch_bounds__(WG) k_signal_max(size_t total, double gamma, nbgas_state s,
                                                   double* __restrict__ block_max) {
    __shared__ double lds[WG / 64];
    const size_t stride = static_cast<size_t>(gridDim.x) * WG;
    double m = -INFINITY;
    for (size_t i = static_cast<size_t>(blockIdx.x) * WG + threadIdx.x; i < total; i += stride) {
        double cs_sq = gamma * s.pressure[i] / s.density[i];
        cs_sq = (s.density[i] > FLOOR) ? cs_sq : 0.0;
        cs_sq = (cs_sq == cs_sq) ? cs_sq : 0.0;
        const double vx = s.velocity_x[i], vy = s.velocity_y[i], vz = s.velocity_z[i];
        const double signal = sqrt(vx * vx + vy * vy + vz * vz) + sqrt(cs_sq);
        m = (signal > m) ? signal : m;      // a NaN signal is skipped, as by Eigen's maxCoeff
    }
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

// out[block] = max of the block's grid-stride share of in[0 .. n)
__global__ void __launch_bounds__(WG) k_max_of(unsigned n, const double* __restrict__ in, double* __restrict__ out) {
    __shared__ double lds[WG / 64];
    double m = -INFINITY;
    for (unsigned i = blockIdx.x * WG + threadIdx.x; i < n; i += gridDim.x * WG) m = (in[i] > m) ? in[i] : m;
    for (int d = 32; d >= 1; d >>= 1) {
        const double o = __shfl_xor(m, d, 64);
        m = (o > m) ? o : m;
    }
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < WG / 64; ++w) m = (lds[w] > m) ? lds[w] : m;
        out[blockIdx.x] = m;
    }
}

// apply_gas_kick (src/integrator.cpp:139-165), primitives included.  WITH_SIGNAL: the per-block
// maxima of k_signal_max as well, from the very pressure and velocities this kernel stores (the
// ones a CFL reduction right after it would read back): one block per WG cells, block_max sized
// to the grid.
template <bool WITH_SIGNAL>
__global__ void __launch_bounds__(WG) k_gas_kick(size_t total, double gamma, nbgas_state s,
                                                 const double* __restrict__ gx, const double* __restrict__ gy,
                                                 const double* __restrict__ gz, double dt, double a3,
                                                 double two_H, double* __restrict__ block_max) {
    __shared__ double lds[WG / 64];
    const size_t idx = static_cast<size_t>(blockIdx.x) * WG + threadIdx.x;
    double signal = -INFINITY;
    if (idx < total) {
        const double gm1 = gamma - 1.0, inv_gm1 = 1.0 / gm1;
        Conserved u;
        u.rho = s.density[idx]; u.m[0] = s.momentum_x[idx]; u.m[1] = s.momentum_y[idx];
        u.m[2] = s.momentum_z[idx]; u.en = s.energy[idx];
        const Cell c = primitives(u, gm1, inv_gm1);
        const double tx = (gx[idx] / a3) - (two_H * c.v[0]);
        const double ty = (gy[idx] / a3) - (two_H * c.v[1]);
        const double tz = (gz[idx] / a3) - (two_H * c.v[2]);
        const double sx = c.rho * tx, sy = c.rho * ty, sz = c.rho * tz;
        const double power = c.v[0] * sx + c.v[1] * sy + c.v[2] * sz;
        s.momentum_x[idx] = c.m[0] + sx * dt;
        s.momentum_y[idx] = c.m[1] + sy * dt;
        s.momentum_z[idx] = c.m[2] + sz * dt;
        s.energy[idx] = c.en + power * dt;
        s.pressure[idx] = c.p;
        s.velocity_x[idx] = c.v[0]; s.velocity_y[idx] = c.v[1]; s.velocity_z[idx] = c.v[2];
        if constexpr (WITH_SIGNAL) {                      // src/gas.cpp:304-318 on the stored values
            double cs_sq = gamma * c.p / c.rho;
            cs_sq = (c.rho > FLOOR) ? cs_sq : 0.0;
            cs_sq = (cs_sq == cs_sq) ? cs_sq : 0.0;
            signal = sqrt(c.v[0] * c.v[0] + c.v[1] * c.v[1] + c.v[2] * c.v[2]) + sqrt(cs_sq);
            signal = (signal == signal) ? signal : -INFINITY;   // a NaN signal is skipped, as by Eigen's maxCoeff
        }
    }
    if constexpr (WITH_SIGNAL) {
        double m = signal;
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

__global__ void __launch_bounds__(WG) k_total_density(size_t total, const double* __restrict__ dm,
                                                      const double* __restrict__ gas, double* __restrict__ out) {
    const size_t idx = static_cast<size_t>(blockIdx.x) * WG + threadIdx.x;
    if (idx < total) out[idx] = dm[idx] + gas[idx];
}

// initialize_gas (src/ics.cpp:256-285), primitives included
__global__ void __launch_bounds__(WG) k_gas_initial(size_t total, double gamma, nbgas_state s,
                                                    const double* __restrict__ dm_density,
                                                    const double* __restrict__ dx, const double* __restrict__ dy,
                                                    const double* __restrict__ dz, double mass_ratio,
                                                    double hubble, double growth_f, int standing,
                                                    double internal_energy) {
    const size_t idx = static_cast<size_t>(blockIdx.x) * WG + threadIdx.x;
    if (idx >= total) return;
    double rho = dm_density[idx] * mass_ratio;
    rho = (rho < FLOOR) ? FLOOR : rho;
    const double vx = standing ? 0.0 : hubble * dx[idx] * growth_f;     // src/ics.cpp:263-268
    const double vy = standing ? 0.0 : hubble * dy[idx] * growth_f;
    const double vz = standing ? 0.0 : hubble * dz[idx] * growth_f;
    Conserved u;
    u.rho = rho;
    u.m[0] = rho * vx; u.m[1] = rho * vy; u.m[2] = rho * vz;
    const double kin = 0.5 * rho * (vx * vx + vy * vy + vz * vz);
    u.en = (rho * internal_energy) + kin;
    const double gm1 = gamma - 1.0, inv_gm1 = 1.0 / gm1;
    const Cell c = primitives(u, gm1, inv_gm1);
    s.density[idx] = c.rho;
    s.momentum_x[idx] = c.m[0]; s.momentum_y[idx] = c.m[1]; s.momentum_z[idx] = c.m[2];
    s.energy[idx] = c.en;
    s.pressure[idx] = c.p;
    s.velocity_x[idx] = c.v[0]; s.velocity_y[idx] = c.v[1]; s.velocity_z[idx] = c.v[2];
}

int check(const nbgas_params* p, const nbgas_state* s) {
    if (!p || !s) return fail(NBGAS_ERR_ARG, "params or state is NULL");
    if (p->mesh_size < 2 || p->mesh_size > 1290)
        return fail(NBGAS_ERR_ARG, "mesh_size must be in [2, 1290] (got %d)", p->mesh_size);
    if (!(p->cell_size > 0.0) || !std::isfinite(p->cell_size)) return fail(NBGAS_ERR_ARG, "cell_size must be positive");
    if (!(p->gamma > 1.0) || !std::isfinite(p->gamma)) return fail(NBGAS_ERR_ARG, "gamma must exceed 1");
    const double* all[9] = {s->density, s->momentum_x, s->momentum_y, s->momentum_z, s->energy,
                            s->pressure, s->velocity_x, s->velocity_y, s->velocity_z};
    for (const double* a : all)
        if (!a) return fail(NBGAS_ERR_ARG, "a mesh of the state is NULL");
    return NBGAS_OK;
}

Fields conserved_of(const nbgas_state* s) {
    return Fields{{s->density, s->momentum_x, s->momentum_y, s->momentum_z, s->energy}};
}

#define ENTER()                                                                   \
    std::lock_guard<std::mutex> lock_(g_mu);                                      \
    if (!g.ready) return fail(NBGAS_ERR_NOINIT, "nbgas_init has not been called"); \
    HIPCHK(hipSetDevice(g.device))

}  // namespace

extern "C" {

int nbgas_init(int device) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g.ready) {
        if (g.device == device) return NBGAS_OK;
        return fail(NBGAS_ERR_ARG, "already initialised on device %d", g.device);
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        (void)hipGetLastError();
        return fail(NBGAS_ERR_HIP, "no HIP device is visible; libnbgas has no CPU path");
    }
    if (device < 0 || device >= count) return fail(NBGAS_ERR_ARG, "device %d out of range [0, %d)", device, count);
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipHostMalloc(&g.pinned, sizeof(double) * PINNED_DOUBLES, hipHostMallocDefault));
    g.device = device;
    g.ready = true;
    return NBGAS_OK;
}

void nbgas_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g.ready) return;
    (void)hipSetDevice(g.device);
    (void)hipDeviceSynchronize();
    g.scratch.release();
    g.blockmax.release();
    if (g.cfl_done) (void)hipEventDestroy(g.cfl_done);
    g.cfl_done = nullptr;
    g.cfl_blocks = 0;
    if (g.pinned) (void)hipHostFree(g.pinned);
    g.pinned = nullptr;
    g.ready = false;
    g.device = -1;
}

const char* nbgas_last_error(void) { return nbc::error_text(); }

int nbgas_update_primitives_device(const nbgas_params* p, const nbgas_state* s, void* stream) {
    ENTER();
    CHK(check(p, s));
    const size_t total = static_cast<size_t>(p->mesh_size) * p->mesh_size * p->mesh_size;
    hipLaunchKernelGGL(k_primitives, dim3(cdiv(total, WG)), dim3(WG), 0, static_cast<hipStream_t>(stream), total,
                       p->gamma, conserved_of(s), *s);
    HIPCHK(hipGetLastError());
    return NBGAS_OK;
}

int nbgas_hydro_step_device(const nbgas_params* p, const nbgas_state* s, double dt, void* stream) {
    ENTER();
    CHK(check(p, s));
    if (!std::isfinite(dt)) return fail(NBGAS_ERR_ARG, "dt must be finite");
    const int N = p->mesh_size;
    const size_t total = static_cast<size_t>(N) * N * N;
    CHK(g.scratch.ensure(5 * total * sizeof(double)));
    Fields mine = conserved_of(s), other;
    for (int c = 0; c < 5; ++c) other.f[c] = g.scratch.as<double>() + c * total;
    const double factor = dt / p->cell_size;                               // src/gas.cpp:251
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(cdiv(total, WG)), wg(WG);
    hipLaunchKernelGGL(k_sweep<0>, grid, wg, 0, st, N, factor, p->gamma, mine, other);
    hipLaunchKernelGGL(k_sweep<1>, grid, wg, 0, st, N, factor, p->gamma, other, mine);
    hipLaunchKernelGGL(k_sweep<2>, grid, wg, 0, st, N, factor, p->gamma, mine, other);
    hipLaunchKernelGGL(k_primitives, grid, wg, 0, st, total, p->gamma, other, *s);   // :298
    HIPCHK(hipGetLastError());
    return NBGAS_OK;
}

int nbgas_cfl_timestep_enqueue_device(const nbgas_params* p, const nbgas_state* s, void* stream) {
    ENTER();
    CHK(check(p, s));
    const size_t total = static_cast<size_t>(p->mesh_size) * p->mesh_size * p->mesh_size;
    const unsigned blocks = std::min<unsigned>(cdiv(total, WG), 1024u);
    CHK(g.blockmax.ensure(sizeof(double) * blocks));
    if (!g.cfl_done) HIPCHK(hipEventCreateWithFlags(&g.cfl_done, hipEventDisableTiming));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_signal_max, dim3(blocks), dim3(WG), 0, st, total, p->gamma, *s, g.blockmax.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(g.pinned, g.blockmax.p, sizeof(double) * blocks, hipMemcpyDeviceToHost, st));   // 1024 doubles fit
    HIPCHK(hipEventRecord(g.cfl_done, st));
    g.cfl_blocks = blocks;
    return NBGAS_OK;
}

int nbgas_cfl_timestep_finish(const nbgas_params* p, double* dt_out) {
    ENTER();
    if (!p || !dt_out) return fail(NBGAS_ERR_ARG, "NULL argument");
    if (g.cfl_blocks == 0) return fail(NBGAS_ERR_ARG, "nbgas_cfl_timestep_enqueue_device has not been called");
    HIPCHK(hipEventSynchronize(g.cfl_done));
    const double* host = static_cast<const double*>(g.pinned);
    double m = host[0];
    for (unsigned b = 1; b < g.cfl_blocks; ++b) m = (host[b] > m) ? host[b] : m;
    g.cfl_blocks = 0;
    if (m < 1e-9) *dt_out = std::numeric_limits<double>::infinity();       // src/gas.cpp:320
    else *dt_out = (p->cell_size / m) * p->cfl_safety_factor;              // :322
    return NBGAS_OK;
}

int nbgas_cfl_timestep_device(const nbgas_params* p, const nbgas_state* s, double* dt_out, void* stream) {
    if (!dt_out) return fail(NBGAS_ERR_ARG, "dt_out is NULL");
    const int rc = nbgas_cfl_timestep_enqueue_device(p, s, stream);
    return rc != NBGAS_OK ? rc : nbgas_cfl_timestep_finish(p, dt_out);
}

int nbgas_kick_device(const nbgas_params* p, const nbgas_state* s, const double* gx, const double* gy,
                      const double* gz, double dt, double scale_factor, double hubble, void* stream) {
    ENTER();
    CHK(check(p, s));
    if (!gx || !gy || !gz) return fail(NBGAS_ERR_ARG, "NULL mesh acceleration");
    const size_t total = static_cast<size_t>(p->mesh_size) * p->mesh_size * p->mesh_size;
    const double a3 = scale_factor * scale_factor * scale_factor;          // src/integrator.cpp:140
    hipLaunchKernelGGL(k_gas_kick<false>, dim3(cdiv(total, WG)), dim3(WG), 0, static_cast<hipStream_t>(stream), total,
                       p->gamma, *s, gx, gy, gz, dt, a3, 2 * hubble, static_cast<double*>(nullptr));
    HIPCHK(hipGetLastError());
    return NBGAS_OK;
}

int nbgas_kick_cfl_enqueue_device(const nbgas_params* p, const nbgas_state* s, const double* gx, const double* gy,
                                  const double* gz, double dt, double scale_factor, double hubble, void* stream) {
    ENTER();
    CHK(check(p, s));
    if (!gx || !gy || !gz) return fail(NBGAS_ERR_ARG, "NULL mesh acceleration");
    const size_t total = static_cast<size_t>(p->mesh_size) * p->mesh_size * p->mesh_size;
    const unsigned blocks = cdiv(total, WG);
    CHK(g.blockmax.ensure(sizeof(double) * (blocks + PINNED_DOUBLES)));
    if (!g.cfl_done) HIPCHK(hipEventCreateWithFlags(&g.cfl_done, hipEventDisableTiming));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double a3 = scale_factor * scale_factor * scale_factor;          // src/integrator.cpp:140
    double* partial = g.blockmax.as<double>();
    hipLaunchKernelGGL(k_gas_kick<true>, dim3(blocks), dim3(WG), 0, st, total, p->gamma, *s, gx, gy, gz, dt, a3,
                       2 * hubble, partial);
    unsigned left = blocks;
    if (left > PINNED_DOUBLES) {   // more partial maxima than the page-locked buffer holds: one more, small pass
        hipLaunchKernelGGL(k_max_of, dim3(PINNED_DOUBLES / 4), dim3(WG), 0, st, left, partial, partial + blocks);
        partial += blocks;
        left = PINNED_DOUBLES / 4;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(g.pinned, partial, sizeof(double) * left, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(g.cfl_done, st));
    g.cfl_blocks = left;
    return NBGAS_OK;
}

int nbgas_total_density_device(const nbgas_params* p, const nbgas_state* s, const double* dm_density,
                               double* total_out, void* stream) {
    ENTER();
    CHK(check(p, s));
    if (!dm_density || !total_out) return fail(NBGAS_ERR_ARG, "NULL mesh");
    const size_t total = static_cast<size_t>(p->mesh_size) * p->mesh_size * p->mesh_size;
    hipLaunchKernelGGL(k_total_density, dim3(cdiv(total, WG)), dim3(WG), 0, static_cast<hipStream_t>(stream), total,
                       dm_density, s->density, total_out);
    HIPCHK(hipGetLastError());
    return NBGAS_OK;
}

int nbgas_initialize_device(const nbgas_params* p, const nbgas_state* s, const double* dm_density,
                            const double* disp_x, const double* disp_y, const double* disp_z,
                            double mass_ratio, double hubble, double growth_f, int standing,
                            double internal_energy, void* stream) {
    ENTER();
    CHK(check(p, s));
    if (!dm_density || !disp_x || !disp_y || !disp_z) return fail(NBGAS_ERR_ARG, "NULL mesh");
    const size_t total = static_cast<size_t>(p->mesh_size) * p->mesh_size * p->mesh_size;
    hipLaunchKernelGGL(k_gas_initial, dim3(cdiv(total, WG)), dim3(WG), 0, static_cast<hipStream_t>(stream), total,
                       p->gamma, *s, dm_density, disp_x, disp_y, disp_z, mass_ratio, hubble, growth_f,
                       standing, internal_energy);
    HIPCHK(hipGetLastError());
    return NBGAS_OK;
}

}  // extern "C"
