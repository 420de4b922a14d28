// nbpp_api.hip - the C ABI of libnbpp (include/nbpp.h): device context, workspace,
// launch logic.  All device code is in nbpp_kernels.hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/nbpp.h"
#include "nbpp_kernels.hip.h"
#include "nbpp_sym.hip.h"

using namespace nbpp;

// Workspace of nbpp_accumulate_device_noalloc, the entry point that must not allocate: one array
// in the code object, so every GPU that loads the library holds its own copy from load time until
// the process ends.  In host code this name is only a handle for hipGetSymbolAddress, never an
// address: nbpp_init asks the runtime where the current device's copy lives and everything else
// (kernel arguments, memsets, copies) goes through that pointer (Ctx::fixed).
static __device__ __align__(256) unsigned char scratch[NBPP_NOALLOC_SCRATCH_BYTES];

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                  \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess)                                                         \
            return fail(e_ == hipErrorOutOfMemory ? NBPP_ERR_NOMEM : NBPP_ERR_HIP,    \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                    \
    } while (0)

#define CHK(expr)                  \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != NBPP_OK) return rc_; \
    } while (0)

// Fixed device memory that buffers are carved from, front to back, instead of being allocated.
struct Arena {
    char* base = nullptr;
    size_t size = 0, used = 0;
};

// Device buffer that only grows.  With an arena it takes its memory from there (a piece it has
// outgrown is not reused before the arena is reset) and never calls the allocator.
struct Buf {
    void* p = nullptr;
    size_t cap = 0;
    Arena* arena = nullptr;
    int ensure(size_t bytes) {
        if (bytes <= cap) return NBPP_OK;
        if (arena) {
            const size_t off = (arena->used + 255) & ~static_cast<size_t>(255);
            if (off > arena->size || bytes > arena->size - off)
                return fail(NBPP_ERR_RANGE,
                            "this call needs more than the %zu bytes of the library's fixed workspace "
                            "(a %zu-byte buffer at offset %zu); use nbpp_accumulate_device",
                            arena->size, bytes, off);
            p = arena->base + off;
            cap = bytes;
            arena->used = off + bytes;
            return NBPP_OK;
        }
        if (p) HIPCHK(hipFree(p));
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 4096;
        HIPCHK(hipMalloc(&p, want));
        cap = want;
        return NBPP_OK;
    }
    void release() {
        if (p && !arena) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// One sorted copy of a particle set.
struct Sorted {
    Buf cell, rank, order, start, x, y, z, gm, box;
    int n = 0, npad = 0;
    template <class F> void each(F f) {
        for (Buf* b : {&cell, &rank, &order, &start, &x, &y, &z, &gm, &box}) f(*b);
    }
    void release() {
        each([](Buf& b) { b.release(); });
    }
};

// Every device buffer a compute call works in.
struct Work {
    Buf count, blocksum;              // binning scratch (shared by source and target sorts)
    Sorted src, tgt;
    Buf part[3];                      // per-slice partial sums
    Buf host_in[4], host_acc[3];      // staging for the host-pointer entry point
    Buf blockmax, counter, massinfo;
    Buf biglist;                      // cells too full for the one-thread sort: two lists of count, ids
    Buf ghostcode;                    // P3M: image code of every sorted entry
    template <class F> void each(F f) {
        for (Buf* b : {&count, &blocksum, &part[0], &part[1], &part[2], &host_in[0],
                       &host_in[1], &host_in[2], &host_in[3], &host_acc[0], &host_acc[1],
                       &host_acc[2], &blockmax, &counter, &massinfo, &biglist, &ghostcode})
            f(*b);
        src.each(f);
        tgt.each(f);
    }
    void release() {
        each([](Buf& b) { b.release(); });
    }
};

struct Ctx : Work {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    void* pinned = nullptr;           // 8 KiB of page-locked host memory: small read-backs land here without staging
    Arena fixed;                      // the device's copy of the fixed workspace (nbpp_accumulate_device_noalloc)
    double* bounce = nullptr;         // page-locked host buffer the sums of nbpp_accumulate come back through
    size_t bounce_bytes = 0;
    double last_occupancy = 0.0;   // clustering measure of the most recent P3M call
    // The pair-once kernel's pre-pass (k_sym_region_max) is not asked for in the calls that follow
    // one whose fullest brick region did not fit, while the particle count and the search grid stay
    // the same (a run clusters further, it does not spread out again): it is tried again every
    // SYM_RETRY_CALLS calls.  The pre-pass sits in front of the call's one read-back.
    int sym_skip_left = 0;
    int sym_skip_n = 0, sym_skip_nside = 0;
    int64_t opt_force_kernel = 0, opt_timing = 0, opt_force_i = 0, opt_force_split = 0,
            opt_p3m_radius = 0, opt_no_uniform = 0, opt_deterministic = 0, opt_p3m_lanes = 0, opt_p3m_pooled = 0, opt_p3m_sym = 0, opt_pool_wg = 0, opt_row_lanes = 0, opt_rows_pooled = 0;
    nbpp_stats stats{};
    // The working buffers are ONE set per process, and a compute call returns while its kernels are
    // still running.  A later call on another stream would overwrite them under those kernels, so every
    // compute call leaves an event behind and the next one, if it comes on a different stream, makes
    // its stream wait for it: calls from several streams are correct, their device work just does
    // not overlap.
    hipEvent_t busy = nullptr;
    hipStream_t busy_stream = nullptr;
    bool busy_recorded = false;
};

Ctx g;
std::mutex g_mu;

inline int cdiv(long long a, long long b) { return static_cast<int>((a + b - 1) / b); }

// What kind of memory a caller's pointer names.  fp64 atomics issued by a kernel are only
// guaranteed to land in ordinary device allocations (hipMalloc and the like); page-locked host
// memory (hipHostMalloc / hipHostRegister) is reachable from kernels by plain loads and stores.
enum class Mem { Device, PinnedHost, Other };
Mem classify(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();   // an address the runtime has never seen
        return Mem::Other;
    }
    if (a.type == hipMemoryTypeDevice) return Mem::Device;
    if (a.type == hipMemoryTypeHost) return Mem::PinnedHost;
    return Mem::Other;             // managed, unregistered, array
}
bool all_device(const double* a, const double* b, const double* c) {
    return classify(a) == Mem::Device && classify(b) == Mem::Device && classify(c) == Mem::Device;
}

int check_params(const nbpp_params* p) {
    if (!p) return fail(NBPP_ERR_ARG, "params is NULL");
    if (!(p->domain_size > 0.0) || !std::isfinite(p->domain_size))
        return fail(NBPP_ERR_ARG, "domain_size must be positive and finite");
    if (!(p->soft_sq >= 0.0)) return fail(NBPP_ERR_ARG, "soft_sq must be >= 0");
    if (!std::isfinite(p->G)) return fail(NBPP_ERR_ARG, "G must be finite");
    if (p->use_pm) {
        // trans_width == 0 (cutoff_transition_width_factor = 0) is legal in the reference: r_switch
        // then equals the cutoff radius and no accepted pair reaches the taper's division
        // pm_soft_sq is (cell / 2)^2 in the reference (src/particles.cpp:146-147), never zero; the
        // pooled kernels evaluate a target's own record and rely on (r^2 + pm_soft_sq)^(-3/2) being
        // finite at r = 0 for that record to add exactly nothing
        if (!(p->cutoff_sq > 0.0) || !(p->trans_width >= 0.0) || !(p->pm_soft_sq > 0.0) ||
            !(p->r_switch >= 0.0))
            return fail(NBPP_ERR_ARG,
                        "use_pm needs cutoff_sq > 0, trans_width >= 0, pm_soft_sq > 0, "
                        "r_switch >= 0");
    }
    return NBPP_OK;
}

PairConsts make_consts(const nbpp_params* p) {
    PairConsts k;
    k.L = p->domain_size;
    k.half_L = 0.5 * p->domain_size;
    k.eps2 = p->soft_sq;
    k.eps_pm2 = p->pm_soft_sq;
    k.r_switch = p->r_switch;
    k.r_switch_sq = p->r_switch_sq;
    k.cutoff_sq = p->cutoff_sq;
    k.inv_width = (p->use_pm && p->trans_width > 0.0) ? 1.0 / p->trans_width : 0.0;
    k.qscale = static_cast<double>(1 << QBITS) / p->domain_size;
    k.c15 = 1.5;
    k.c1875 = 1.875;
    return k;
}

// Members of every cell into ascending entry order (k_cell_canon, k_cell_canon_mid, k_cell_canon_big).
int canonical_order(int ncell, int n_entries, const int* start, int* order, int* tmp_val, int* tmp_rank,
                    hipStream_t s) {
    const size_t n_mid = static_cast<size_t>(n_entries) / (CANON_TINY + 1) + 2;
    const size_t n_big = static_cast<size_t>(n_entries) / (CANON_WAVE + 1) + 2;
    CHK(g.biglist.ensure(sizeof(int) * (n_mid + n_big)));
    int* mid = g.biglist.as<int>();
    int* big = mid + n_mid;
    HIPCHK(hipMemsetAsync(mid, 0, sizeof(int), s));
    HIPCHK(hipMemsetAsync(big, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_cell_canon, dim3(cdiv(ncell, WG)), dim3(WG), 0, s, ncell, start, order, mid, big);
    hipLaunchKernelGGL(k_cell_canon_mid, dim3(2048), dim3(WG), 0, s, start, order, mid);
    hipLaunchKernelGGL(k_cell_canon_big, dim3(1024), dim3(WG), 0, s, start, order, big, tmp_val, tmp_rank);
    HIPCHK(hipGetLastError());
    return NBPP_OK;
}

// Exclusive scan of count[0..n) into start[0..n); n includes the trailing zero entry, so
// start[n-1] is the grand total.
int exclusive_scan(int n, const int* count, int* start, hipStream_t s) {
    const int nblocks = cdiv(n, SCAN_BLOCK);
    CHK(g.blocksum.ensure(sizeof(int) * (nblocks + 1)));
    int* bs = g.blocksum.as<int>();
    hipLaunchKernelGGL(k_scan_reduce, dim3(nblocks), dim3(WG), 0, s, n, count, bs);
    hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(WG), 0, s, nblocks, bs, bs + nblocks);
    hipLaunchKernelGGL(k_scan_apply, dim3(nblocks), dim3(WG), 0, s, n, count, start, bs);
    HIPCHK(hipGetLastError());
    return NBPP_OK;
}

// Counting sort of n particles by cell of `grid`; leaves out.order (sorted -> original) and
// out.start, the cell offsets (ncell + 1 entries).
int sort_by_cell(int n, const double* x, const double* y, const double* z, const BinGrid& grid,
                 Sorted& out, hipStream_t s) {
    const long long ncell = 1LL * grid.nside * grid.nside * grid.nside;
    if (ncell + 1 > 0x7fffffffLL) return fail(NBPP_ERR_RANGE, "bin grid too large");
    CHK(g.count.ensure(sizeof(int) * (ncell + 1)));
    CHK(out.start.ensure(sizeof(int) * (ncell + 1)));
    CHK(out.cell.ensure(sizeof(int) * n));
    CHK(out.rank.ensure(sizeof(int) * n));
    CHK(out.order.ensure(sizeof(int) * n));
    HIPCHK(hipMemsetAsync(g.count.p, 0, sizeof(int) * (ncell + 1), s));
    hipLaunchKernelGGL(k_bin_count, dim3(cdiv(n, WG)), dim3(WG), 0, s, n, x, y, z, grid,
                       out.cell.as<int>(), out.rank.as<int>(), g.count.as<int>());
    CHK(exclusive_scan(static_cast<int>(ncell + 1), g.count.as<int>(), out.start.as<int>(), s));
    hipLaunchKernelGGL(k_bin_scatter, dim3(cdiv(n, WG)), dim3(WG), 0, s, n, out.cell.as<int>(),
                       out.rank.as<int>(), out.start.as<int>(), out.order.as<int>());
    // cell / rank have done their job in the scatter: scratch for the sort of over-full cells
    CHK(canonical_order(static_cast<int>(ncell), n, out.start.as<int>(), out.order.as<int>(),
                        out.cell.as<int>(), out.rank.as<int>(), s));
    out.n = n;
    return NBPP_OK;
}

// Z-order grid for the direct kernel: power-of-two side, about 8 particles per cell.
BinGrid direct_grid(int n, double L) {
    int nside = 1;
    while (nside < 1024 && 8.0 * (2.0 * nside) * (2.0 * nside) * (2.0 * nside) <= n) nside *= 2;
    BinGrid gr;
    gr.nside = nside;
    gr.scale = nside / L;
    gr.morton = 1;
    return gr;
}

template <int MODE, bool SOFT0>
int launch_pairs_i(int I, dim3 grid, const PairArgs& a, hipStream_t s) {
    switch (I) {
        case 1: hipLaunchKernelGGL((k_pairs<1, MODE, SOFT0>), grid, dim3(WG), 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_pairs<2, MODE, SOFT0>), grid, dim3(WG), 0, s, a); break;
        default: hipLaunchKernelGGL((k_pairs<4, MODE, SOFT0>), grid, dim3(WG), 0, s, a); break;
    }
    HIPCHK(hipGetLastError());
    return NBPP_OK;
}

int launch_pairs(int mode, bool soft0, int I, dim3 grid, const PairArgs& a, hipStream_t s) {
    if (mode == 2) return launch_pairs_i<2, true>(I, grid, a, s);
    if (mode == 1)
        return soft0 ? launch_pairs_i<1, true>(I, grid, a, s) : launch_pairs_i<1, false>(I, grid, a, s);
    return soft0 ? launch_pairs_i<0, true>(I, grid, a, s) : launch_pairs_i<0, false>(I, grid, a, s);
}

// All-pairs path (direct summation, or the P3M pair function over all pairs).
int run_allpairs(const nbpp_params* p, int variant, int n_src, const double* x, const double* y,
                 const double* z, const double* m, int tgt_begin, int n_tgt, double* ax,
                 double* ay, double* az, hipStream_t s, int sym_part = 0, int sym_parts = 1) {
    const PairConsts k = make_consts(p);
    const BinGrid grid = direct_grid(n_src, p->domain_size);

    // sources: sort, gather, boxes
    CHK(sort_by_cell(n_src, x, y, z, grid, g.src, s));
    const int npad = cdiv(n_src, TJ) * TJ;
    g.src.npad = npad;
    for (Buf* b : {&g.src.x, &g.src.y, &g.src.z, &g.src.gm}) CHK(b->ensure(sizeof(double) * npad));
    CHK(g.src.box.ensure(sizeof(int) * 8 * (npad / SUB)));
    // equal masses (every dark-matter particle of the reference has DM_PARTICLE_MASS,
    // src/ics.cpp:209-212) are detected on the device; the pair kernel then skips the per-pair
    // mass multiply
    CHK(g.massinfo.ensure(16));
    int* uniform_flag = g.massinfo.as<int>();
    double* common_gm = reinterpret_cast<double*>(g.massinfo.as<char>() + 8);
    hipLaunchKernelGGL(k_fill_int, dim3(1), dim3(64), 0, s, 1, uniform_flag,
                       (variant == NBPP_KERNEL_P3M_ALLPAIRS || g.opt_no_uniform) ? 0 : 1);
    hipLaunchKernelGGL(k_mass_uniform, dim3(cdiv(n_src, WG)), dim3(WG), 0, s, n_src, m, p->G,
                       uniform_flag, common_gm);
    hipLaunchKernelGGL(k_gather_sources, dim3(npad / WG), dim3(WG), 0, s, n_src, npad,
                       g.src.order.as<int>(), x, y, z, m, p->G, uniform_flag, g.src.x.as<double>(),
                       g.src.y.as<double>(), g.src.z.as<double>(), g.src.gm.as<double>());
    hipLaunchKernelGGL(k_source_boxes, dim3(npad / WG), dim3(WG), 0, s, npad,
                       g.src.x.as<double>(), g.src.y.as<double>(), g.src.z.as<double>(), k.qscale,
                       g.src.box.as<int>());
    HIPCHK(hipGetLastError());

    if (variant == NBPP_KERNEL_DIRECT_SYMMETRIC) {
        if (g.opt_timing) {
            CHK(g.counter.ensure(2 * sizeof(unsigned long long)));
            HIPCHK(hipMemsetAsync(g.counter.p, 0, 2 * sizeof(unsigned long long), s));
            HIPCHK(hipEventRecord(g.ev[1], s));
        }
        // four particles per lane (fewest rotations per pair) unless there are too few particles
        // to form groups of 256
        int I = n_src >= (1 << 16) ? 4 : (n_src >= (1 << 14) ? 2 : 1);
        if (g.opt_force_i == 1 || g.opt_force_i == 2 || g.opt_force_i == 4) I = static_cast<int>(g.opt_force_i);
        SymArgs a{};
        a.sx = g.src.x.as<double>(); a.sy = g.src.y.as<double>(); a.sz = g.src.z.as<double>();
        a.sgm = g.src.gm.as<double>(); a.sbox = g.src.box.as<int>(); a.order = g.src.order.as<int>();
        a.n = n_src; a.n_tiles = npad / 64;
        const int NG = a.n_tiles / I;
        const int parts = std::max(1, sym_parts), part = std::min(sym_part, parts - 1);
        // every group's visit list is cut into parts * local slices: `parts` shares for the ranks of
        // a multi-GPU run, and enough local slices (blockIdx.y) for ~16 wavefronts per SIMD-slot
        const int visits_per_group = std::max(1, ((NG - 1) / 2) * I);
        int local = std::max(1, cdiv(16384, NG));
        if (g.opt_force_split > 0) local = static_cast<int>(g.opt_force_split);
        local = std::max(1, std::min(local, visits_per_group / parts));
        a.n_slices = parts * local;
        a.slice_begin = part * local;
        a.ax = ax; a.ay = ay; a.az = az;
        a.uniform_mass = uniform_flag; a.common_gm = common_gm;
        a.tile_counts = g.opt_timing ? g.counter.as<unsigned long long>() : nullptr;
        a.k = k;
        const dim3 grid(cdiv(NG, WG / 64), local);
        const bool soft0 = p->soft_sq == 0.0;
        if (I == 1) { if (soft0) hipLaunchKernelGGL((k_pairs_sym<1, true>), grid, dim3(WG), 0, s, a); else hipLaunchKernelGGL((k_pairs_sym<1, false>), grid, dim3(WG), 0, s, a); }
        else if (I == 2) { if (soft0) hipLaunchKernelGGL((k_pairs_sym<2, true>), grid, dim3(WG), 0, s, a); else hipLaunchKernelGGL((k_pairs_sym<2, false>), grid, dim3(WG), 0, s, a); }
        else { if (soft0) hipLaunchKernelGGL((k_pairs_sym<4, true>), grid, dim3(WG), 0, s, a); else hipLaunchKernelGGL((k_pairs_sym<4, false>), grid, dim3(WG), 0, s, a); }
        HIPCHK(hipGetLastError());
        g.stats.pairs_evaluated = static_cast<uint64_t>(n_src) * static_cast<uint64_t>(n_src - 1) / parts;
        g.stats.pair_launches = 1;
        return NBPP_OK;
    }

    // targets: the sources themselves, or a separately sorted slice
    const double *tx, *ty, *tz;
    const int* tperm;
    if (tgt_begin == 0 && n_tgt == n_src) {
        tx = g.src.x.as<double>(); ty = g.src.y.as<double>(); tz = g.src.z.as<double>();
        tperm = g.src.order.as<int>();
    } else {
        CHK(sort_by_cell(n_tgt, x + tgt_begin, y + tgt_begin, z + tgt_begin, grid, g.tgt, s));
        for (Buf* b : {&g.tgt.x, &g.tgt.y, &g.tgt.z}) CHK(b->ensure(sizeof(double) * n_tgt));
        hipLaunchKernelGGL(k_gather_targets, dim3(cdiv(n_tgt, WG)), dim3(WG), 0, s, n_tgt, n_tgt,
                           g.tgt.order.as<int>(), x + tgt_begin, y + tgt_begin, z + tgt_begin,
                           g.tgt.x.as<double>(), g.tgt.y.as<double>(), g.tgt.z.as<double>());
        HIPCHK(hipGetLastError());
        tx = g.tgt.x.as<double>(); ty = g.tgt.y.as<double>(); tz = g.tgt.z.as<double>();
        tperm = g.tgt.order.as<int>();
    }
    if (g.opt_timing) HIPCHK(hipEventRecord(g.ev[1], s));

    // launch shape: I targets per lane, source range split over blockIdx.y when the target
    // count alone cannot fill 256 CUs
    // two targets per lane measured fastest at 128^3 (no spills at 4 waves/SIMD, fewer straddling
    // sub-tiles than four); one per lane below 2^17 targets to keep more workgroups in flight
    int I = n_tgt >= (1 << 17) ? 2 : 1;
    if (g.opt_force_i == 1 || g.opt_force_i == 2 || g.opt_force_i == 4) I = static_cast<int>(g.opt_force_i);
    const int blocks_i = cdiv(n_tgt, WG * I);
    const int n_tiles = npad / TJ;
    int split = std::min(n_tiles, std::max(1, cdiv(1024, blocks_i)));
    if (g.opt_force_split > 0) split = std::min<int>(n_tiles, static_cast<int>(g.opt_force_split));
    const int tiles_per_split = cdiv(n_tiles, split);
    split = cdiv(n_tiles, tiles_per_split);

    PairArgs a{};
    a.sx = g.src.x.as<double>(); a.sy = g.src.y.as<double>(); a.sz = g.src.z.as<double>();
    a.sgm = g.src.gm.as<double>(); a.sbox = g.src.box.as<int>(); a.n_src_pad = npad;
    a.tx = tx; a.ty = ty; a.tz = tz; a.tperm = tperm; a.n_tgt = n_tgt;
    a.tiles_per_split = tiles_per_split;
    a.ax = ax; a.ay = ay; a.az = az;
    a.n_tgt_pad = cdiv(n_tgt, 64) * 64;
    a.uniform_mass = uniform_flag;
    a.common_gm = common_gm;
    a.k = k;
    if (g.opt_timing) {
        CHK(g.counter.ensure(2 * sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync(g.counter.p, 0, 2 * sizeof(unsigned long long), s));
        a.tile_counts = g.counter.as<unsigned long long>();
    }
    if (split > 1) {
        for (Buf& b : g.part) CHK(b.ensure(sizeof(double) * static_cast<size_t>(split) * a.n_tgt_pad));
        a.px = g.part[0].as<double>(); a.py = g.part[1].as<double>(); a.pz = g.part[2].as<double>();
    }
    const int mode = variant == NBPP_KERNEL_P3M_ALLPAIRS ? 2 : (variant == NBPP_KERNEL_DIRECT_WRAP ? 1 : 0);
    CHK(launch_pairs(mode, p->soft_sq == 0.0, I, dim3(blocks_i, split), a, s));
    if (split > 1) {
        hipLaunchKernelGGL(k_reduce_partials, dim3(cdiv(n_tgt, WG)), dim3(WG), 0, s, n_tgt,
                           a.n_tgt_pad, split, a.px, a.py, a.pz, tperm, ax, ay, az);
        HIPCHK(hipGetLastError());
    }
    g.stats.pairs_evaluated = static_cast<uint64_t>(n_tgt) * static_cast<uint64_t>(n_src - 1);
    g.stats.pair_launches = 1;
    return NBPP_OK;
}

// Search grid of the P3M cell kernel: `radius` cells cover the cutoff radius with a 1e-9
// relative margin (binning rounds at the 1e-16 level).  Returns false when the box holds
// fewer than 2*radius + 3 cells per side: then the wrap of a cell index no longer implies
// the minimum image and the all-pairs kernel is used instead.
// Radius 2 means 25 runs of 5 larger cells per target instead of 49 runs of 7: 22 % more
// candidates, but longer runs, which the lanes of a wavefront walk with less imbalance and less
// round-up to whole trips.  Measured faster from 48^3 particles up (256^3 lattice 5.07 against
// 5.37 ms, the clustered 128^3 set 2.05 against 2.36 ms); small clustered sets, which need
// several lanes per target, do better with the 49 shorter runs (32^3 at a = 1: 0.21 against
// 0.25 ms).
bool p3m_search_grid(const nbpp_params* p, int n_src, int* radius, int* nside) {
#ifdef NBPP_EXPERIMENT_SKIN_PERCENT   // cells laid out for a cutoff radius this much larger (what a list kept for several steps would need)
    const double rc = std::sqrt(p->cutoff_sq) * (1.0 + 1e-9) * (1.0 + NBPP_EXPERIMENT_SKIN_PERCENT / 100.0);
#else
    const double rc = std::sqrt(p->cutoff_sq) * (1.0 + 1e-9);
#endif
    int R = g.opt_p3m_radius > 0 ? static_cast<int>(g.opt_p3m_radius) : (n_src >= (1 << 17) ? 2 : 3);
    R = std::min(R, 8);
    const double cells = std::floor(R * p->domain_size / rc);
    if (!(cells >= 2 * R + 3)) return false;
    int n = static_cast<int>(std::min(cells, 1290.0));  // 1290^3 < 2^31
    *radius = R;
    *nside = n;
    return true;
}

// Single-precision r^2 bound of the cutoff pre-filter.  Stored coordinates lie in
// [0, X], X = L (1 + (R+1)/nside), so rounding one to fp32 moves it by at most 2^-24 X; a
// displacement component of a pair inside the cutoff is therefore off by at most
// e = 2^-23 X + 2^-23 rc (two roundings plus the subtraction's own), and the fp32 sum of
// squares, three roundings deep, by a factor below 1 + 2^-22.  Hence for r <= rc
//   r2_fp32 <= (rc^2 + 2 sqrt(3) e rc + 3 e^2) (1 + 2^-22);
// the bound below adds slack and rounds up.  Candidates it lets through are re-tested in fp64.
float p3m_prefilter_bound(const nbpp_params* p, const GhostGrid& grid) {
    const double rc = std::sqrt(p->cutoff_sq);
    const double X = p->domain_size * (1.0 + (grid.R + 1.0) / grid.nside);
    const double e = std::ldexp(X, -23) + std::ldexp(rc, -23);
    const double bound = (rc * rc + 2.0 * std::sqrt(3.0) * e * rc + 3.0 * e * e) * (1.0 + std::ldexp(1.0, -20));
    return std::nextafter(static_cast<float>(bound), INFINITY);
}

int run_p3m_cells(const nbpp_params* p, int n_src, const double* x, const double* y,
                  const double* z, const double* m, int tgt_begin, int n_tgt, double* ax,
                  double* ay, double* az, hipStream_t s) {
    const PairConsts k = make_consts(p);
    int R = 0, nside = 0;
    if (!p3m_search_grid(p, n_src, &R, &nside)) return fail(NBPP_ERR_ARG, "box too small for the cell search");
    if (n_src > (1 << 30))   // real + ghost entries are counted in 32-bit integers
        return fail(NBPP_ERR_RANGE, "the P3M cell kernel takes at most 2^30 sources (got %d)", n_src);
    GhostGrid grid;
    grid.nside = nside;
    grid.R = R;
    grid.np = nside + 2 * R;
    grid.scale = nside / p->domain_size;
    grid.L = p->domain_size;
    grid.keep_lo = 0;
    grid.keep_len = nside;
    const long long ncell = 1LL * grid.np * grid.np * grid.np;
    if (ncell + 1 > 0x7fffffffLL) return fail(NBPP_ERR_RANGE, "search grid too large");
    const bool all_targets = tgt_begin == 0 && n_tgt == n_src;

    // 0. a slice of the particles as targets (one rank's share of a multi-GPU run): if the slice
    // occupies only part of the box in x, only that slab and a halo of R cells need binning.
    // The occupied x cells are read back (nside integers); the slab is the complement of their
    // largest empty circular gap.
    if (!all_targets && n_tgt > 0) {
        CHK(g.blocksum.ensure(sizeof(int) * nside));
        int* hist = g.blocksum.as<int>();
        HIPCHK(hipMemsetAsync(hist, 0, sizeof(int) * nside, s));
        hipLaunchKernelGGL(k_target_x_cells, dim3(cdiv(n_tgt, WG)), dim3(WG), 0, s, n_tgt, x + tgt_begin, grid, hist);
        std::vector<int> occupied(nside);
        HIPCHK(hipMemcpyAsync(occupied.data(), hist, sizeof(int) * nside, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        int best_start = 0, best_len = 0;      // longest run of empty cells on the circle
        for (int c = 0; c < nside; ++c) {
            if (occupied[c] == 0) continue;    // a run of empty cells starts after an occupied one
            int len = 0;
            while (len < nside && occupied[(c + 1 + len) % nside] == 0) ++len;
            if (len > best_len) { best_len = len; best_start = (c + 1) % nside; }
        }
        if (best_len > 2 * R) {
            grid.keep_lo = (best_start + best_len - R) % nside;        // R cells before the first occupied one
            grid.keep_len = nside - best_len + 2 * R;
        }
    }

    // 1. histogram over real + ghost cells, offsets
    CHK(g.count.ensure(sizeof(int) * (ncell + 1)));
    CHK(g.src.start.ensure(sizeof(int) * (ncell + 1)));
    int* count = g.count.as<int>();
    int* start = g.src.start.as<int>();
    HIPCHK(hipMemsetAsync(count, 0, sizeof(int) * (ncell + 1), s));
    CHK(g.tgt.rank.ensure(sizeof(int) * n_src));
    int* rank = g.tgt.rank.as<int>();
    hipLaunchKernelGGL(k_ghost_count, dim3(cdiv(n_src, WG)), dim3(WG), 0, s, n_src, x, y, z, grid,
                       count, rank);
    CHK(exclusive_scan(static_cast<int>(ncell + 1), count, start, s));
    // the one host read-back of the call: real + ghost entries and the clustering measure of
    // k_occupancy (the mean, over entries, of the occupancy of the entry's own cell: 1 + the mean
    // cell occupancy for evenly spread particles, tens to thousands once they have clustered),
    // which decides how many lanes share a target and whether the filter's survivors are pooled
    int n_entries = 0;
    double occupancy = 1.0;
    // bricks of the pair-once kernel (k_p3m_sym) and the fullest brick's region
    const SymBrick brick = sym_brick(R);
    const int nbx = cdiv(nside, brick.bx), nby = cdiv(nside, brick.by), nbz = cdiv(nside, brick.bz);
    unsigned long long region_max = ~0ull;
    {
        CHK(g.counter.ensure(4 * sizeof(unsigned long long)));
        unsigned long long* occ = g.counter.as<unsigned long long>();
        HIPCHK(hipMemsetAsync(occ, 0, 4 * sizeof(unsigned long long), s));
        hipLaunchKernelGGL(k_occupancy, dim3(static_cast<int>(std::min<long long>(256, cdiv(ncell, WG)))), dim3(WG), 0, s,
                           static_cast<int>(ncell), count, start, occ);
        // (not asked for by itself while the last call found the particles clustered: it would not be
        // chosen, and its pre-pass costs 30-50 us)
        if (g.sym_skip_n != n_src || g.sym_skip_nside != nside) g.sym_skip_left = 0;
        const bool sym_rested = g.opt_p3m_sym == 0 && g.sym_skip_left > 0;
        if (sym_rested) --g.sym_skip_left;
        // (nor by itself below 2^16 targets, which always get several lanes each: see `lanes` below)
        const bool sym_possible = all_targets && sym_brick_fits(brick, R) && !sym_rested &&
                                  (g.opt_p3m_sym > 0 ||
                                   (g.opt_p3m_sym == 0 && g.last_occupancy < 6.5 && n_tgt >= 65536 && !g.opt_deterministic &&
                                    (g.opt_p3m_lanes == 0 || g.opt_p3m_lanes == 1)));
        if (sym_possible)
            hipLaunchKernelGGL(k_sym_region_max, dim3(cdiv(1LL * nbx * nby * nbz, WG)), dim3(WG), 0, s, grid, start,
                               brick, nbx, nby, nbz, occ + 2);
        unsigned long long* host = static_cast<unsigned long long*>(g.pinned);
        HIPCHK(hipMemcpyAsync(host, occ, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(hipEventRecord(g.ev[3], s));
        // Work that needs nothing of what is read back is queued behind the copy, so the GPU has
        // something to do while the host wakes up, decides and launches the rest: when every
        // particle is a target the list of targets is the real entries in sorted order, one
        // contiguous piece per column of the grid, and follows from the cell offsets alone.
        if (all_targets && n_tgt > 0) {
            CHK(g.tgt.order.ensure(sizeof(int) * n_tgt));
            const int ncol = grid.np * grid.np;
            CHK(g.src.cell.ensure(sizeof(int) * (ncol + 1)));
            CHK(g.src.rank.ensure(sizeof(int) * (ncol + 1)));
            int* col_count = g.src.cell.as<int>();
            int* col_start = g.src.rank.as<int>();
            if (ncol <= COLUMN_STARTS_SMALL_MAX) {
                hipLaunchKernelGGL(k_column_starts_small, dim3(1), dim3(1024), 0, s, grid, start, col_start);
            } else {
                hipLaunchKernelGGL(k_column_counts, dim3(cdiv(ncol + 1, WG)), dim3(WG), 0, s, grid, start, col_count);
                CHK(exclusive_scan(ncol + 1, col_count, col_start, s));
            }
            hipLaunchKernelGGL(k_column_targets, dim3(std::min(ncol, 65536)), dim3(WG), 0, s, grid, start, col_start,
                               g.tgt.order.as<int>());
        }
        HIPCHK(hipEventSynchronize(g.ev[3]));
        n_entries = static_cast<int>(host[1]);
        if (n_entries > 0) occupancy = static_cast<double>(host[0]) / n_entries;
        if (sym_possible) {
            region_max = host[2];
            if (region_max > static_cast<unsigned long long>(SYM_CAP)) {
                g.sym_skip_left = SYM_RETRY_CALLS - 1;
                g.sym_skip_n = n_src;
                g.sym_skip_nside = nside;
            }
        }
        g.stats.cell_occupancy = occupancy;
        g.last_occupancy = occupancy;
    }
    if (n_entries < (grid.keep_len == nside ? n_src : n_tgt))
        return fail(NBPP_ERR_HIP, "binning lost particles (%d entries)", n_entries);

    // Which pair kernel will run is known now (everything it depends on has been read back), and
    // the binning below depends on it: only k_p3m_cells needs the members of a cell in a fixed
    // order.
    const bool wide = n_entries >= (1 << 27);   // 32-byte records beyond 32-bit byte offsets
    // lanes per target: more of them while the launch is too small to occupy the GPU.  Lanes of
    // one target walk different rows, which costs the coalescing that neighbouring targets on the
    // same row have, so on evenly spread particles one lane per target wins from about 2^16
    // targets up (measured at 32^3 to 128^3: 32^3 best with 2-4 lanes, 48^3 and larger with 1).
    // Clustered sets have a few very long targets and want more lanes for the same count
    // (32^3 at a = 0.5 to 1: 8 lanes), hence the occupancy factor.
    int lanes = 1;
    const double enough = 65536.0 * std::min(std::max(occupancy, 1.0), 16.0);
    while (lanes < 8 && static_cast<double>(n_tgt) * lanes < enough) lanes *= 2;
    if (g.opt_p3m_lanes == 1 || g.opt_p3m_lanes == 2 || g.opt_p3m_lanes == 4 || g.opt_p3m_lanes == 8 ||
        g.opt_p3m_lanes == 16)
        lanes = static_cast<int>(g.opt_p3m_lanes);
    // every unordered pair once, reactions gathered per brick in LDS (k_p3m_sym): a full
    // evaluation whose fullest brick region fits, sums not asked to be reproducible, outputs in
    // ordinary device memory (fp64 atomics).  Chosen by itself for large, even sets (one lane per
    // target, occupancy measure below the pooled kernel's threshold): measured against
    // k_p3m_cells on jittered lattices, pair kernel alone, 0.125 / 0.131 ms at 64^3,
    // 0.673 / 0.711 at 128^3, 2.05 / 2.30 at 192^3, 4.67 / 5.00 at 256^3.
    const bool table_fits = static_cast<unsigned long long>(ncell) <= START_TABLE_CHECKED_MAX_CELLS;
    // G lanes of a target walking every row together (k_p3m_rows) instead of several lanes of a
    // target owning whole rows each: every call that would get more than one lane per target
    int row_lanes = 0;
    if (!wide && table_fits) {
        if (g.opt_row_lanes > 0) row_lanes = static_cast<int>(g.opt_row_lanes);
        else if (g.opt_row_lanes == 0 && g.opt_p3m_lanes == 0 && lanes > 1) row_lanes = P3M_ROWS_LANES;
        // ... and large sets once nearly every row a target meets is long (the 400-blobs set, occupancy
        // measure 442: 10.3 ms with 4 lanes per target against 12.4 with the pooled kernel)
        else if (g.opt_row_lanes == 0 && g.opt_p3m_lanes == 0 && g.opt_p3m_pooled <= 0 &&
                 occupancy >= P3M_ROWS_LARGE_MIN_OCCUPANCY)
            row_lanes = P3M_ROWS_LARGE_LANES;
    }
    // Row walk with the pooled drain (k_p3m_rows_pool): forced (NBPP_OPT_P3M_ROWS_POOLED = 1 with a
    // lane count from NBPP_OPT_P3M_ROW_LANES), or by itself with two lanes per target for sets that
    // fill the GPU with one, from a clustering measure of P3M_ROWS_POOL_MIN_OCCUPANCY up to where
    // the plain row walk takes over.  Measured on the 128^3 run, pair kernel alone, against the
    // pooled kernel: a = 0.21 (2.8) 0.832 / 0.846 ms, a = 0.37 (3.8) 1.02 / 1.09, a = 0.51 (7.3) 1.33 /
    // 1.41, a = 0.74 (19) 2.12 / 2.58, a = 0.89 (30) 2.75 / 3.51 (plain row walk, 4 lanes: 3.02); four
    // lanes per target 1.49 / 2.25 / 2.79 at the last three, eight lose (few target slots, so the
    // drain's LDS adds collide).
    const bool rows_pool_can = !wide && table_fits && !g.opt_deterministic &&
                               n_entries + P3M_POOL_TRIP < (1 << P3M_POOL_LANE_SHIFT);
    bool rows_pool = rows_pool_can && g.opt_rows_pooled > 0 && row_lanes >= 2 && row_lanes <= 16;
    const bool sym = row_lanes == 0 && all_targets && region_max <= static_cast<unsigned long long>(SYM_CAP) && !wide && table_fits &&
                     n_entries + SYM_TRIP < (1 << SYM_JBITS) && all_device(ax, ay, az) &&
                     (g.opt_p3m_sym > 0 ||
                      (g.opt_p3m_sym == 0 && !g.opt_deterministic && lanes == 1 && occupancy < 6.0));
    // one lane per target: the survivors of the filter are pooled per wavefront and evaluated 64
    // at a time whatever target they belong to (k_p3m_pool), unless reproducible sums are asked
    // for.  Measured, kernel time pooled against per lane: 256^3 jittered lattice (occupancy
    // measure 2.6) 4.79 against 5.08 ms; the 128^3 run at a = 0.25 (2.9) 0.93 against 1.00 ms, at
    // a = 0.33 (3.4) 1.07 against 1.22, at a = 0.40 (4.4) 1.23 against 1.47, at a = 0.51 (7.3) 1.40
    // against 1.96; 400 tight blobs 12.5 against 14.8 ms.
    if (!sym && rows_pool_can && g.opt_rows_pooled == 0 && row_lanes == 0 && g.opt_row_lanes == 0 &&
        g.opt_p3m_lanes == 0 && g.opt_p3m_pooled == 0 && lanes == 1 && occupancy >= P3M_ROWS_POOL_MIN_OCCUPANCY) {
        rows_pool = true;      // where the pair-once kernel does not take the call
        row_lanes = 2;
    }
    const bool pooled = row_lanes == 0 && lanes == 1 &&
                        (g.opt_p3m_pooled > 0 || (g.opt_p3m_pooled == 0 && !g.opt_deterministic));

    // 2. scatter and records.  k_p3m_cells promises sums that repeat from run to run: the members
    // of every cell are put into ascending original order before their records are gathered.  The
    // other two kernels add in whatever order the hardware takes their atomics, so for them the
    // order the scatter's own atomics produced will do (three kernels and two fills less: 0.15 of
    // 0.38 ms of binning on the clustered 128^3 set, 0.1 of 0.9 ms on the 256^3 lattice).  (Writing
    // the records from the scatter itself, without a gather, was measured too: the scattered
    // 4-byte stores cost more than the gather's reads, 0.87 against 0.81 ms at 256^3.)
    const bool fixed_order = !(sym || pooled || (row_lanes > 0 && !g.opt_deterministic));
    CHK(g.src.order.ensure(sizeof(int) * n_entries));
    CHK(g.src.x.ensure(sizeof(double4) * n_entries));  // AoS records live in the x buffer
    int* entries = g.src.order.as<int>();
    CHK(g.ghostcode.ensure(static_cast<size_t>(n_entries)));
    uint8_t* codes = g.ghostcode.as<uint8_t>();
    double4* rec = g.src.x.as<double4>();
    // and their fp32 copies, one array per axis, in the y buffer
    const size_t f_stride = (static_cast<size_t>(n_entries) + std::max({P3M_TRIP, P3M_POOL_TRIP, SYM_TRIP}) + 63) / 64 * 64;
    CHK(g.src.y.ensure(sizeof(float) * 3 * f_stride));
    float* fx = g.src.y.as<float>();
    float* fy = fx + f_stride;
    float* fz = fy + f_stride;
    hipLaunchKernelGGL(k_ghost_scatter, dim3(cdiv(n_src, WG)), dim3(WG), 0, s, n_src, x, y, z, grid,
                       start, count, rank, entries, codes);
    if (fixed_order) {
        CHK(g.src.cell.ensure(sizeof(int) * (n_entries + 1)));   // scratch here, target flags / slots below
        CHK(g.src.rank.ensure(sizeof(int) * (n_entries + 1)));
        CHK(canonical_order(static_cast<int>(ncell), n_entries, start, entries, g.src.cell.as<int>(),
                            g.src.rank.as<int>(), s));
    }
    // k_p3m_sym takes the candidates' positions as 16-bit integers instead (SymQuant): the same buffer,
    // x | y << 16 in the first n_entries + 8 words, z in half words behind them.  The range covers the
    // ghost layers and half a cell more; two positions r apart differ by at most r * scale + sqrt(3).
    SymQuant quant{0.0, 1.0};
    uint32_t* qxy = reinterpret_cast<uint32_t*>(fx);
    uint16_t* qz = reinterpret_cast<uint16_t*>(qxy + f_stride);
    int qbound = 0;
    if (sym) {
        const double cell = p->domain_size / nside, margin = (R + 0.5) * cell;
        quant.lo = -margin;
        quant.scale = 65535.0 / (p->domain_size + 2.0 * margin);
        const double reach = std::sqrt(p->cutoff_sq) * quant.scale + 1.7320508075688773 + 1e-6;
        qbound = static_cast<int>(std::floor(reach * reach)) + 1;
        hipLaunchKernelGGL(k_ghost_gather<true>, dim3(cdiv(n_entries, WG)), dim3(WG), 0, s, n_entries, entries, codes,
                           x, y, z, m, p->G, p->domain_size, rec, fx, fy, fz, quant, qxy, qz);
    } else {
        hipLaunchKernelGGL(k_ghost_gather<false>, dim3(cdiv(n_entries, WG)), dim3(WG), 0, s, n_entries, entries, codes,
                           x, y, z, m, p->G, p->domain_size, rec, fx, fy, fz, quant, qxy, qz);
    }

    // 3. list of targets in sorted order (every particle a target: built in front of the read-back, above)
    CHK(g.tgt.order.ensure(sizeof(int) * n_tgt));
    int* tlist = g.tgt.order.as<int>();
    if (!all_targets) {
        // a slice of the particles: flag the real entries in it and compact
        CHK(g.src.cell.ensure(sizeof(int) * (n_entries + 1)));
        CHK(g.src.rank.ensure(sizeof(int) * (n_entries + 1)));
        int* flag = g.src.cell.as<int>();
        int* slot = g.src.rank.as<int>();
        hipLaunchKernelGGL(k_target_flags, dim3(cdiv(n_entries + 1, WG)), dim3(WG), 0, s, n_entries, entries, codes,
                           tgt_begin, n_tgt, flag);
        CHK(exclusive_scan(n_entries + 1, flag, slot, s));
        hipLaunchKernelGGL(k_target_scatter, dim3(cdiv(n_entries, WG)), dim3(WG), 0, s, n_entries, flag, slot,
                           tlist);
    }
    HIPCHK(hipGetLastError());
    constexpr size_t counter_bytes = sizeof(unsigned long long) * P3M_COUNT_STRIPES * P3M_COUNT_STRIDE;
    CHK(g.counter.ensure(counter_bytes));
    if (g.opt_timing) {   // the accepted-pair counter is only kept when timing
        HIPCHK(hipMemsetAsync(g.counter.p, 0, counter_bytes, s));
        HIPCHK(hipEventRecord(g.ev[1], s));
    }

    // 4. pairs
    P3MArgs a{};
    a.rec = rec; a.fx = fx; a.fy = fy; a.fz = fz; a.prefilter_sq = p3m_prefilter_bound(p, grid);
    a.qxy = qxy; a.qz = qz; a.qbound = qbound;
    a.entries = entries; a.cell_start = start; a.tlist = tlist;
    a.n_entries = n_entries; a.n_tgt = n_tgt; a.tgt_begin = tgt_begin; a.g = grid;
    a.ax = ax; a.ay = ay; a.az = az;
    a.accepted = g.opt_timing ? g.counter.as<unsigned long long>() : nullptr;
    a.k = k;
    const dim3 grid_dim(cdiv(static_cast<long long>(n_tgt) * lanes, P3M_CELLS_WG)), wg_dim(P3M_CELLS_WG);
    const bool soft0 = p->soft_sq == 0.0;
    g.stats.p3m_pooled = 0;
    if (sym) {
        const dim3 bricks(nbx * nby * nbz);
        if (soft0) hipLaunchKernelGGL((k_p3m_sym<true>), bricks, dim3(SYM_WG), 0, s, a, brick, nbx, nby, nbz);
        else hipLaunchKernelGGL((k_p3m_sym<false>), bricks, dim3(SYM_WG), 0, s, a, brick, nbx, nby, nbz);
        HIPCHK(hipGetLastError());
        g.stats.p3m_pooled = 2;
        g.stats.pairs_evaluated = 0;
        g.stats.pair_launches = 1;
        return NBPP_OK;
    }
    if (rows_pool) {
        const dim3 rows_grid(cdiv(static_cast<long long>(n_tgt) * row_lanes, 64)), rows_wg(64);
#define NBPP_LAUNCH_ROWS_POOL(S0)                                                                         \
    switch (row_lanes) {                                                                                  \
        case 2: hipLaunchKernelGGL((k_p3m_rows_pool<S0, 2>), rows_grid, rows_wg, 0, s, a); break;         \
        case 4: hipLaunchKernelGGL((k_p3m_rows_pool<S0, 4>), rows_grid, rows_wg, 0, s, a); break;         \
        case 8: hipLaunchKernelGGL((k_p3m_rows_pool<S0, 8>), rows_grid, rows_wg, 0, s, a); break;         \
        default: hipLaunchKernelGGL((k_p3m_rows_pool<S0, 16>), rows_grid, rows_wg, 0, s, a); break;       \
    }
        if (soft0) { NBPP_LAUNCH_ROWS_POOL(true) } else { NBPP_LAUNCH_ROWS_POOL(false) }
#undef NBPP_LAUNCH_ROWS_POOL
        HIPCHK(hipGetLastError());
        g.stats.p3m_pooled = 4;
        g.stats.pairs_evaluated = 0;
        g.stats.pair_launches = 1;
        return NBPP_OK;
    }
    if (row_lanes > 0) {
        const dim3 rows_grid(cdiv(static_cast<long long>(n_tgt) * row_lanes, 64)), rows_wg(64);
#define NBPP_LAUNCH_ROWS(S0)                                                                         \
    switch (row_lanes) {                                                                             \
        case 2: hipLaunchKernelGGL((k_p3m_rows<S0, 2>), rows_grid, rows_wg, 0, s, a); break;         \
        case 4: hipLaunchKernelGGL((k_p3m_rows<S0, 4>), rows_grid, rows_wg, 0, s, a); break;         \
        case 8: hipLaunchKernelGGL((k_p3m_rows<S0, 8>), rows_grid, rows_wg, 0, s, a); break;         \
        case 16: hipLaunchKernelGGL((k_p3m_rows<S0, 16>), rows_grid, rows_wg, 0, s, a); break;       \
        case 32: hipLaunchKernelGGL((k_p3m_rows<S0, 32>), rows_grid, rows_wg, 0, s, a); break;       \
        default: hipLaunchKernelGGL((k_p3m_rows<S0, 64>), rows_grid, rows_wg, 0, s, a); break;       \
    }
        if (soft0) { NBPP_LAUNCH_ROWS(true) } else { NBPP_LAUNCH_ROWS(false) }
#undef NBPP_LAUNCH_ROWS
        HIPCHK(hipGetLastError());
        g.stats.p3m_pooled = 3;
        g.stats.pairs_evaluated = 0;
        g.stats.pair_launches = 1;
        return NBPP_OK;
    }
    g.stats.p3m_pooled = pooled ? 1 : 0;
    if (pooled) {
        // its ring entries keep the sorted position in 26 bits; beyond that the two-word form
        const bool pool_wide = wide || !table_fits || n_entries + P3M_POOL_TRIP >= (1 << P3M_POOL_LANE_SHIFT);
        const int pwg = g.opt_pool_wg ? static_cast<int>(g.opt_pool_wg) : 64;
        const dim3 pool_grid(cdiv(n_tgt, pwg)), pool_wg(pwg);
#define NBPP_LAUNCH_POOL(S0, W)                                                                         \
    switch (pwg) {                                                                                      \
        case 64: hipLaunchKernelGGL((k_p3m_pool<S0, W, 64>), pool_grid, pool_wg, 0, s, a); break;       \
        case 128: hipLaunchKernelGGL((k_p3m_pool<S0, W, 128>), pool_grid, pool_wg, 0, s, a); break;     \
        default: hipLaunchKernelGGL((k_p3m_pool<S0, W, 256>), pool_grid, pool_wg, 0, s, a); break;      \
    }
        if (soft0) {
            if (pool_wide) { NBPP_LAUNCH_POOL(true, true) } else { NBPP_LAUNCH_POOL(true, false) }
        } else {
            if (pool_wide) { NBPP_LAUNCH_POOL(false, true) } else { NBPP_LAUNCH_POOL(false, false) }
        }
#undef NBPP_LAUNCH_POOL
        HIPCHK(hipGetLastError());
        g.stats.pairs_evaluated = 0;
        g.stats.pair_launches = 1;
        return NBPP_OK;
    }
#define NBPP_LAUNCH_P3M(S0, W, L) hipLaunchKernelGGL((k_p3m_cells<S0, W, L>), grid_dim, wg_dim, 0, s, a)
#define NBPP_P3M_LANES(S0, W)                                       \
    switch (lanes) {                                                \
        case 1: NBPP_LAUNCH_P3M(S0, W, 1); break;                   \
        case 2: NBPP_LAUNCH_P3M(S0, W, 2); break;                   \
        case 4: NBPP_LAUNCH_P3M(S0, W, 4); break;                   \
        case 8: NBPP_LAUNCH_P3M(S0, W, 8); break;                   \
        default: NBPP_LAUNCH_P3M(S0, W, 16); break;                 \
    }
    if (soft0) {
        if (wide) { NBPP_P3M_LANES(true, true) } else { NBPP_P3M_LANES(true, false) }
    } else {
        if (wide) { NBPP_P3M_LANES(false, true) } else { NBPP_P3M_LANES(false, false) }
    }
#undef NBPP_P3M_LANES
#undef NBPP_LAUNCH_P3M
    HIPCHK(hipGetLastError());
    g.stats.pairs_evaluated = 0;  // candidates depend on the data; see pairs_accepted
    g.stats.pair_launches = 1;
    return NBPP_OK;
}

int compute_in_order(const nbpp_params* p, size_t n_src, const double* x, const double* y,
                     const double* z, const double* m, size_t tgt_begin, size_t n_tgt,
                     double* ax, double* ay, double* az, hipStream_t s, int sym_part, int sym_parts);

// Every compute entry point ends up here, holding g_mu.
int compute_on_device(const nbpp_params* p, size_t n_src, const double* x, const double* y,
                      const double* z, const double* m, size_t tgt_begin, size_t n_tgt,
                      double* ax, double* ay, double* az, hipStream_t s, int sym_part = -1,
                      int sym_parts = 1) {
    if (n_src > 0x7fffff00ULL) return fail(NBPP_ERR_RANGE, "n_src exceeds 2^31 - 256");
    if (g.busy_recorded && g.busy_stream != s) HIPCHK(hipStreamWaitEvent(s, g.busy, 0));
    const int rc = compute_in_order(p, n_src, x, y, z, m, tgt_begin, n_tgt, ax, ay, az, s, sym_part, sym_parts);
    // (also after a failure: whatever was queued before it may still be running)
    if (hipEventRecord(g.busy, s) == hipSuccess) {
        g.busy_stream = s;
        g.busy_recorded = true;
    } else {
        (void)hipGetLastError();
        if (rc == NBPP_OK) return fail(NBPP_ERR_HIP, "hipEventRecord failed after the compute call");
    }
    return rc;
}

int compute_in_order(const nbpp_params* p, size_t n_src, const double* x, const double* y,
                     const double* z, const double* m, size_t tgt_begin, size_t n_tgt,
                     double* ax, double* ay, double* az, hipStream_t s, int sym_part, int sym_parts) {
    g.stats = nbpp_stats{};
    g.stats.n_sources = n_src;
    g.stats.n_targets = n_tgt;
    int variant;
    if (p->use_pm) {
        variant = NBPP_KERNEL_P3M_CELLS;
        int radius, nside;
        if (g.opt_force_kernel == NBPP_KERNEL_P3M_ALLPAIRS || n_src < 2048 ||
            !p3m_search_grid(p, static_cast<int>(n_src), &radius, &nside))
            variant = NBPP_KERNEL_P3M_ALLPAIRS;
        if (g.opt_force_kernel == NBPP_KERNEL_P3M_CELLS && p3m_search_grid(p, static_cast<int>(n_src), &radius, &nside))
            variant = NBPP_KERNEL_P3M_CELLS;
    } else {
        variant = NBPP_KERNEL_DIRECT;
        if (g.opt_force_kernel == NBPP_KERNEL_DIRECT_WRAP) variant = NBPP_KERNEL_DIRECT_WRAP;
        const bool all_targets = tgt_begin == 0 && n_tgt == n_src;
        // each unordered pair once: chosen by itself for full evaluations of 2^14 particles or
        // more (measured faster than the ordered kernel from 32^3 up, equal at 16^3)
        if (all_targets && (g.opt_force_kernel == NBPP_KERNEL_DIRECT_SYMMETRIC ||
                            (g.opt_force_kernel == 0 && !g.opt_deterministic && n_src >= (1u << 14))))
            variant = NBPP_KERNEL_DIRECT_SYMMETRIC;
        if (sym_part >= 0) variant = NBPP_KERNEL_DIRECT_SYMMETRIC;
        // that kernel adds to the outputs with fp64 atomics: only for ordinary device memory
        if (variant == NBPP_KERNEL_DIRECT_SYMMETRIC && !all_device(ax, ay, az)) {
            if (sym_part >= 0 || g.opt_force_kernel == NBPP_KERNEL_DIRECT_SYMMETRIC)
                return fail(NBPP_ERR_ARG,
                            "the pair-once kernel adds with fp64 atomics and needs ax, ay, az in ordinary "
                            "device memory (hipMalloc), not host-mapped or managed memory");
            variant = NBPP_KERNEL_DIRECT;
        }
    }
    g.stats.kernel_variant = variant;
    if (g.opt_timing) HIPCHK(hipEventRecord(g.ev[0], s));
    if (variant == NBPP_KERNEL_P3M_CELLS) {
        CHK(run_p3m_cells(p, static_cast<int>(n_src), x, y, z, m, static_cast<int>(tgt_begin),
                          static_cast<int>(n_tgt), ax, ay, az, s));
    } else {
        CHK(run_allpairs(p, variant, static_cast<int>(n_src), x, y, z, m,
                         static_cast<int>(tgt_begin), static_cast<int>(n_tgt), ax, ay, az, s,
                         std::max(sym_part, 0), sym_parts));
    }
    if (g.opt_timing) {
        HIPCHK(hipEventRecord(g.ev[2], s));
        HIPCHK(hipEventSynchronize(g.ev[2]));
        float a = 0, b = 0;
        HIPCHK(hipEventElapsedTime(&a, g.ev[0], g.ev[1]));
        HIPCHK(hipEventElapsedTime(&b, g.ev[1], g.ev[2]));
        g.stats.prepare_ms = a;
        g.stats.pair_kernel_ms = b;
        g.stats.total_ms = a + b;
        if (variant == NBPP_KERNEL_P3M_CELLS) {
            unsigned long long stripes[P3M_COUNT_STRIPES * P3M_COUNT_STRIDE];
            HIPCHK(hipMemcpy(stripes, g.counter.p, sizeof(stripes), hipMemcpyDeviceToHost));
            unsigned long long hits = 0;
            for (int i = 0; i < P3M_COUNT_STRIPES; ++i) hits += stripes[i * P3M_COUNT_STRIDE];
            g.stats.pairs_accepted = hits;
#ifdef NBPP_SYM_STATS
            unsigned long long d[8] = {};
            for (int i = 0; i < P3M_COUNT_STRIPES; ++i)
                for (int w = 1; w < 8; ++w) d[w] += stripes[i * P3M_COUNT_STRIDE + w];
            if (d[7])
                std::fprintf(stderr, "sym stats: waves %llu, per wave: trips %.1f rows %.1f drain pair-steps %.1f (live lanes %.1f%%), "
                             "per target: candidates %.1f survivors %.1f\n", d[7], double(d[1]) / d[7], double(d[2]) / d[7],
                             double(d[3]) / d[7], 100.0 * d[4] / (64.0 * d[3]), double(d[5]) / (64.0 * d[7]), double(d[6]) / (64.0 * d[7]));
#endif
        } else {
            unsigned long long tc[2] = {0, 0};
            HIPCHK(hipMemcpy(tc, g.counter.p, sizeof(tc), hipMemcpyDeviceToHost));
            g.stats.subtiles_total = tc[0];
            g.stats.subtiles_wrapped = tc[1];
            g.stats.pairs_accepted = g.stats.pairs_evaluated;
        }
    }
    return NBPP_OK;
}

}  // namespace

extern "C" {

const char* nbpp_last_error(void) { return g_err; }

int nbpp_init(int device) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g.ready) return NBPP_OK;
    int count = 0;
    HIPCHK(hipGetDeviceCount(&count));
    if (count <= 0) return fail(NBPP_ERR_HIP, "no HIP device visible");
    if (device < 0) HIPCHK(hipGetDevice(&device));
    if (device >= count) return fail(NBPP_ERR_ARG, "device %d out of range (%d visible)", device, count);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(NBPP_ERR_HIP, "device %d is %s; libnbpp is built for gfx950 only", device,
                    prop.gcnArchName);
    HIPCHK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    for (auto& e : g.ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipEventCreateWithFlags(&g.busy, hipEventDisableTiming));
    g.busy_recorded = false;
    HIPCHK(hipHostMalloc(&g.pinned, 8192, hipHostMallocDefault));
    // this device's copy of the fixed workspace (the device was made current above)
    void* fixed = nullptr;
    HIPCHK(hipGetSymbolAddress(&fixed, HIP_SYMBOL(scratch)));
    g.fixed.base = static_cast<char*>(fixed);
    g.fixed.size = NBPP_NOALLOC_SCRATCH_BYTES;
    g.fixed.used = 0;
    g.device = device;
    g.ready = true;
    return NBPP_OK;
}

void nbpp_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g.ready) return;
    (void)hipSetDevice(g.device);
    (void)hipStreamSynchronize(g.stream);
    g.release();
    g.fixed = Arena{};
    for (auto& e : g.ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    if (g.busy) (void)hipEventDestroy(g.busy);
    g.busy = nullptr;
    g.busy_recorded = false;
    (void)hipStreamDestroy(g.stream);
    g.stream = nullptr;
    if (g.pinned) (void)hipHostFree(g.pinned);
    g.pinned = nullptr;
    if (g.bounce) (void)hipHostFree(g.bounce);
    g.bounce = nullptr;
    g.bounce_bytes = 0;
    g.ready = false;
}

int nbpp_set_option(int key, int64_t value) {
    std::lock_guard<std::mutex> lock(g_mu);   // options are read by compute calls, which hold the same lock
    switch (key) {
        case NBPP_OPT_FORCE_KERNEL: g.opt_force_kernel = value; return NBPP_OK;
        case NBPP_OPT_TIMING: g.opt_timing = value; return NBPP_OK;
        case NBPP_OPT_TARGETS_PER_LANE: g.opt_force_i = value; return NBPP_OK;
        case NBPP_OPT_SOURCE_SPLIT: g.opt_force_split = value; return NBPP_OK;
        case NBPP_OPT_P3M_SEARCH_RADIUS: g.opt_p3m_radius = value; return NBPP_OK;
        case NBPP_OPT_NO_EQUAL_MASS_PATH: g.opt_no_uniform = value; return NBPP_OK;
        case NBPP_OPT_DETERMINISTIC: g.opt_deterministic = value; return NBPP_OK;
        case NBPP_OPT_P3M_LANES_PER_TARGET: g.opt_p3m_lanes = value; return NBPP_OK;
        case NBPP_OPT_P3M_POOLED_DRAIN: g.opt_p3m_pooled = value; return NBPP_OK;
        case NBPP_OPT_P3M_PAIR_ONCE: g.opt_p3m_sym = value; return NBPP_OK;
        case NBPP_OPT_P3M_ROW_LANES:
            if (value != 0 && value != -1 && value != 2 && value != 4 && value != 8 && value != 16 && value != 32 && value != 64)
                return fail(NBPP_ERR_ARG, "NBPP_OPT_P3M_ROW_LANES takes 0, -1, 2, 4, 8, 16, 32 or 64");
            g.opt_row_lanes = value;
            return NBPP_OK;
        case NBPP_OPT_P3M_ROWS_POOLED: g.opt_rows_pooled = value; return NBPP_OK;
        case NBPP_OPT_P3M_POOL_WORKGROUP:
            if (value != 0 && value != 64 && value != 128 && value != 256)
                return fail(NBPP_ERR_ARG, "NBPP_OPT_P3M_POOL_WORKGROUP takes 0, 64, 128 or 256");
            g.opt_pool_wg = value;
            return NBPP_OK;
        default: return fail(NBPP_ERR_ARG, "unknown option key %d", key);
    }
}

int nbpp_get_stats(nbpp_stats* out) {
    if (!out) return fail(NBPP_ERR_ARG, "out is NULL");
    std::lock_guard<std::mutex> lock(g_mu);
    *out = g.stats;
    return NBPP_OK;
}

int nbpp_accumulate_device(const nbpp_params* p, size_t n_src, const double* x, const double* y,
                           const double* z, const double* m, size_t tgt_begin, size_t n_tgt,
                           double* ax, double* ay, double* az, void* stream) {
    if (!g.ready) return fail(NBPP_ERR_NOINIT, "nbpp_init has not been called");
    CHK(check_params(p));
    if (tgt_begin > n_src || n_tgt > n_src - tgt_begin)
        return fail(NBPP_ERR_ARG, "target range [%zu, %zu) is outside the %zu sources", tgt_begin,
                    tgt_begin + n_tgt, n_src);
    if (n_tgt == 0) return NBPP_OK;
    if (!x || !y || !z || !m || !ax || !ay || !az) return fail(NBPP_ERR_ARG, "NULL array pointer");
    std::lock_guard<std::mutex> lock(g_mu);
    HIPCHK(hipSetDevice(g.device));
    hipStream_t s = static_cast<hipStream_t>(stream);   // NULL: the HIP null stream
    return compute_on_device(p, n_src, x, y, z, m, tgt_begin, n_tgt, ax, ay, az, s);
}

int nbpp_accumulate_device_noalloc(const nbpp_params* p, size_t n, const double* x, const double* y,
                                   const double* z, const double* m, double* ax, double* ay,
                                   double* az, void* stream) {
    if (!g.ready) return fail(NBPP_ERR_NOINIT, "nbpp_init has not been called");
    CHK(check_params(p));
    if (n > NBPP_NOALLOC_MAX_N)
        return fail(NBPP_ERR_RANGE,
                    "nbpp_accumulate_device_noalloc takes at most %d particles (got %zu); "
                    "use nbpp_accumulate_device",
                    NBPP_NOALLOC_MAX_N, n);
    if (n == 0) return NBPP_OK;
    if (!x || !y || !z || !m || !ax || !ay || !az) return fail(NBPP_ERR_ARG, "NULL array pointer");
    std::lock_guard<std::mutex> lock(g_mu);
    HIPCHK(hipSetDevice(g.device));
    hipStream_t s = static_cast<hipStream_t>(stream);   // NULL: the HIP null stream
    // For the length of this call the working buffers are empty ones that take their memory from
    // the fixed workspace, front to back; the allocated buffers of the other entry points are set
    // aside and come back untouched.  Every kernel that expects zeros (histograms, counters,
    // list heads) clears them itself, as it must for reused allocated buffers too, so what the
    // last call left in the workspace does not matter.
    Work saved;
    std::swap(static_cast<Work&>(g), saved);
    g.fixed.used = 0;
    static_cast<Work&>(g).each([](Buf& b) { b.arena = &g.fixed; });
    const int rc = compute_on_device(p, n, x, y, z, m, 0, n, ax, ay, az, s);
    static_cast<Work&>(g) = saved;
    return rc;
}

int nbpp_accumulate_pairs_device(const nbpp_params* p, size_t n, const double* x, const double* y,
                                 const double* z, const double* m, int part, int n_parts,
                                 double* ax, double* ay, double* az, void* stream) {
    if (!g.ready) return fail(NBPP_ERR_NOINIT, "nbpp_init has not been called");
    CHK(check_params(p));
    if (p->use_pm) return fail(NBPP_ERR_ARG, "nbpp_accumulate_pairs_device is for use_pm == 0 only");
    if (n_parts < 1 || part < 0 || part >= n_parts)
        return fail(NBPP_ERR_ARG, "part %d is not in [0, %d)", part, n_parts);
    if (n == 0) return NBPP_OK;
    if (!x || !y || !z || !m || !ax || !ay || !az) return fail(NBPP_ERR_ARG, "NULL array pointer");
    std::lock_guard<std::mutex> lock(g_mu);
    HIPCHK(hipSetDevice(g.device));
    hipStream_t s = static_cast<hipStream_t>(stream);   // NULL: the HIP null stream
    return compute_on_device(p, n, x, y, z, m, 0, n, ax, ay, az, s, part, n_parts);
}

void* nbpp_host_alloc(size_t bytes) {
    if (!g.ready) {
        (void)fail(NBPP_ERR_NOINIT, "nbpp_init has not been called");
        return nullptr;
    }
    void* p = nullptr;
    if (bytes == 0 || hipSetDevice(g.device) != hipSuccess ||
        hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        (void)fail(NBPP_ERR_NOMEM, "cannot allocate %zu bytes of page-locked host memory", bytes);
        return nullptr;
    }
    return p;
}

void nbpp_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int nbpp_accumulate(const nbpp_params* p, size_t n, const double* x, const double* y,
                    const double* z, const double* m, double* ax, double* ay, double* az) {
    if (!g.ready) return fail(NBPP_ERR_NOINIT, "nbpp_init has not been called");
    CHK(check_params(p));
    if (n == 0) return NBPP_OK;
    if (!x || !y || !z || !m || !ax || !ay || !az) return fail(NBPP_ERR_ARG, "NULL array pointer");
    std::lock_guard<std::mutex> lock(g_mu);
    HIPCHK(hipSetDevice(g.device));
    const size_t bytes = sizeof(double) * n;
    const double* hin[4] = {x, y, z, m};
    double* hout[3] = {ax, ay, az};
    for (int c = 0; c < 4; ++c) CHK(g.host_in[c].ensure(bytes));
    for (int c = 0; c < 3; ++c) CHK(g.host_acc[c].ensure(bytes));
    // page-locked accumulators (nbpp_host_alloc, hipHostMalloc): the device adds its sums to
    // them in place, through the bus; pageable ones: the sums come back through a page-locked
    // buffer of the library's own and host threads add them
    bool acc_pinned = true;
    for (double* o : hout) acc_pinned = acc_pinned && classify(o) == Mem::PinnedHost;
    auto t0 = std::chrono::steady_clock::now();
    // (from page-locked caller memory these are plain DMA transfers; from pageable memory the
    // runtime stages them)
    for (int c = 0; c < 4; ++c)
        HIPCHK(hipMemcpyAsync(g.host_in[c].p, hin[c], bytes, hipMemcpyHostToDevice, g.stream));
    // the device sums start from zero and are added to the caller's accumulators once, which
    // is the same single rounding as the reference's acc[i] += local (src/particles.cpp:362-364)
    for (int c = 0; c < 3; ++c) HIPCHK(hipMemsetAsync(g.host_acc[c].p, 0, bytes, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    auto t1 = std::chrono::steady_clock::now();
    int rc = compute_on_device(p, n, g.host_in[0].as<double>(), g.host_in[1].as<double>(),
                               g.host_in[2].as<double>(), g.host_in[3].as<double>(), 0, n,
                               g.host_acc[0].as<double>(), g.host_acc[1].as<double>(),
                               g.host_acc[2].as<double>(), g.stream);
    if (rc != NBPP_OK) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    auto t2 = std::chrono::steady_clock::now();
    if (acc_pinned) {
        double* dev_view[3];
        for (int c = 0; c < 3; ++c)
            HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&dev_view[c]), hout[c], 0));
        const int blocks = std::min(4096, cdiv(static_cast<long long>(n), WG));
        hipLaunchKernelGGL(k_add<false>, dim3(blocks), dim3(WG), 0, g.stream, n, dev_view[0], dev_view[1], dev_view[2],
                           g.host_acc[0].as<double>(), g.host_acc[1].as<double>(), g.host_acc[2].as<double>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(g.stream));
    } else {
        if (g.bounce_bytes < 3 * bytes) {
            if (g.bounce) HIPCHK(hipHostFree(g.bounce));
            g.bounce = nullptr;
            g.bounce_bytes = 0;
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&g.bounce), 3 * bytes + 3 * bytes / 8, hipHostMallocDefault));
            g.bounce_bytes = 3 * bytes + 3 * bytes / 8;
        }
        // one transfer per component; while component c + 1 is on the bus, host threads add c
        for (int c = 0; c < 3; ++c) {
            HIPCHK(hipMemcpyAsync(g.bounce + c * n, g.host_acc[c].p, bytes, hipMemcpyDeviceToHost, g.stream));
            HIPCHK(hipEventRecord(g.ev[c], g.stream));
        }
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const size_t nthreads = std::max<size_t>(1, std::min<size_t>({8, hw, n / 65536 + 1}));
        for (int c = 0; c < 3; ++c) {
            HIPCHK(hipEventSynchronize(g.ev[c]));
            double* o = hout[c];
            const double* t = g.bounce + c * n;
            auto add = [o, t](size_t b, size_t e) { for (size_t i = b; i < e; ++i) o[i] += t[i]; };
            std::vector<std::thread> pool;
            const size_t chunk = (n + nthreads - 1) / nthreads;
            for (size_t k = 1; k < nthreads; ++k)
                pool.emplace_back(add, std::min(n, k * chunk), std::min(n, (k + 1) * chunk));
            add(0, std::min(n, chunk));
            for (auto& th : pool) th.join();
        }
    }
    auto t3 = std::chrono::steady_clock::now();
    g.stats.h2d_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    g.stats.d2h_ms = std::chrono::duration<double, std::milli>(t3 - t2).count();
    if (!g.opt_timing) g.stats.total_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    return NBPP_OK;
}

namespace {
int add_arrays(bool clear, size_t n, double* ax, double* ay, double* az, double* bx, double* by, double* bz,
               void* stream) {
    if (!g.ready) return fail(NBPP_ERR_NOINIT, "nbpp_init has not been called");
    if (n == 0) return NBPP_OK;
    if (!ax || !ay || !az || !bx || !by || !bz) return fail(NBPP_ERR_ARG, "NULL array pointer");
    HIPCHK(hipSetDevice(g.device));
    hipStream_t s = static_cast<hipStream_t>(stream);   // NULL: the HIP null stream
    const int blocks = std::min(2048, cdiv(static_cast<long long>(n), WG));
    if (clear) hipLaunchKernelGGL(k_add<true>, dim3(blocks), dim3(WG), 0, s, n, ax, ay, az, bx, by, bz);
    else hipLaunchKernelGGL(k_add<false>, dim3(blocks), dim3(WG), 0, s, n, ax, ay, az, bx, by, bz);
    HIPCHK(hipGetLastError());
    return NBPP_OK;
}
}  // namespace

int nbpp_add_device(size_t n, double* ax, double* ay, double* az, const double* bx,
                    const double* by, const double* bz, void* stream) {
    return add_arrays(false, n, ax, ay, az, const_cast<double*>(bx), const_cast<double*>(by),
                      const_cast<double*>(bz), stream);
}

int nbpp_add_clear_device(size_t n, double* ax, double* ay, double* az, double* bx, double* by,
                          double* bz, void* stream) {
    return add_arrays(true, n, ax, ay, az, bx, by, bz, stream);
}

int nbpp_kick_device(size_t n, double* vx, double* vy, double* vz, const double* ax,
                     const double* ay, const double* az, double dt, double scale_factor,
                     double hubble, void* stream) {
    if (!g.ready) return fail(NBPP_ERR_NOINIT, "nbpp_init has not been called");
    if (n == 0) return NBPP_OK;
    if (!vx || !vy || !vz || !ax || !ay || !az) return fail(NBPP_ERR_ARG, "NULL array pointer");
    HIPCHK(hipSetDevice(g.device));
    hipStream_t s = static_cast<hipStream_t>(stream);   // NULL: the HIP null stream
    const double a3 = scale_factor * scale_factor * scale_factor;  // src/integrator.cpp:169
    const double dt_over_a3 = dt / a3;                             // :170
    const double dt_2H = dt * 2.0 * hubble;                        // :171
    const int blocks = std::min(2048, cdiv(static_cast<long long>(n), WG));
    hipLaunchKernelGGL(k_kick, dim3(blocks), dim3(WG), 0, s, n, vx, vy, vz, ax, ay, az, dt_over_a3,
                       dt_2H);
    HIPCHK(hipGetLastError());
    return NBPP_OK;
}

int nbpp_drift_device(size_t n, double* x, double* y, double* z, const double* vx,
                      const double* vy, const double* vz, double dt, double domain_size,
                      void* stream) {
    if (!g.ready) return fail(NBPP_ERR_NOINIT, "nbpp_init has not been called");
    if (n == 0) return NBPP_OK;
    if (!x || !y || !z || !vx || !vy || !vz) return fail(NBPP_ERR_ARG, "NULL array pointer");
    if (!(domain_size > 0.0)) return fail(NBPP_ERR_ARG, "domain_size must be positive");
    HIPCHK(hipSetDevice(g.device));
    hipStream_t s = static_cast<hipStream_t>(stream);   // NULL: the HIP null stream
    const int blocks = std::min(2048, cdiv(static_cast<long long>(n), WG));
    hipLaunchKernelGGL(k_drift, dim3(blocks), dim3(WG), 0, s, n, x, y, z, vx, vy, vz, dt,
                       domain_size, 1.0 / domain_size);  // inv_domain: src/integrator.cpp:184
    HIPCHK(hipGetLastError());
    return NBPP_OK;
}

int nbpp_kick_drift_device(size_t n, double* x, double* y, double* z, double* vx, double* vy, double* vz,
                           const double* ax, const double* ay, const double* az, double dt_kick,
                           double dt_drift, double scale_factor, double hubble, double domain_size,
                           void* stream) {
    if (!g.ready) return fail(NBPP_ERR_NOINIT, "nbpp_init has not been called");
    if (n == 0) return NBPP_OK;
    if (!x || !y || !z || !vx || !vy || !vz || !ax || !ay || !az) return fail(NBPP_ERR_ARG, "NULL array pointer");
    if (!(domain_size > 0.0)) return fail(NBPP_ERR_ARG, "domain_size must be positive");
    HIPCHK(hipSetDevice(g.device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double a3 = scale_factor * scale_factor * scale_factor;  // src/integrator.cpp:169
    const int blocks = std::min(2048, cdiv(static_cast<long long>(n), WG));
    hipLaunchKernelGGL(k_kick_drift, dim3(blocks), dim3(WG), 0, s, n, x, y, z, vx, vy, vz, ax, ay, az,
                       dt_kick / a3, dt_kick * 2.0 * hubble, dt_drift, domain_size, 1.0 / domain_size);
    HIPCHK(hipGetLastError());
    return NBPP_OK;
}

int nbpp_add_clear_kick_device(size_t n, double* ax, double* ay, double* az, double* bx, double* by,
                               double* bz, double* vx, double* vy, double* vz, double dt,
                               double scale_factor, double hubble, void* stream) {
    if (!g.ready) return fail(NBPP_ERR_NOINIT, "nbpp_init has not been called");
    if (n == 0) return NBPP_OK;
    if (!ax || !ay || !az || !bx || !by || !bz || !vx || !vy || !vz) return fail(NBPP_ERR_ARG, "NULL array pointer");
    HIPCHK(hipSetDevice(g.device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double a3 = scale_factor * scale_factor * scale_factor;
    const int blocks = std::min(2048, cdiv(static_cast<long long>(n), WG));
    hipLaunchKernelGGL(k_add_clear_kick<false>, dim3(blocks), dim3(WG), 0, s, n, ax, ay, az, bx, by, bz, vx, vy, vz,
                       dt / a3, dt * 2.0 * hubble, static_cast<double*>(nullptr));
    HIPCHK(hipGetLastError());
    return NBPP_OK;
}

int nbpp_add_clear_kick_max_device(size_t n, double* ax, double* ay, double* az, double* bx, double* by,
                                   double* bz, double* vx, double* vy, double* vz, double dt,
                                   double scale_factor, double hubble, double* out_host, void* stream) {
    if (!g.ready) return fail(NBPP_ERR_NOINIT, "nbpp_init has not been called");
    if (!out_host) return fail(NBPP_ERR_ARG, "out_host is NULL");
    *out_host = 1e-9;                                        // src/integrator.cpp:202
    if (n == 0) return NBPP_OK;
    if (!ax || !ay || !az || !bx || !by || !bz || !vx || !vy || !vz) return fail(NBPP_ERR_ARG, "NULL array pointer");
    std::lock_guard<std::mutex> lock(g_mu);
    HIPCHK(hipSetDevice(g.device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double a3 = scale_factor * scale_factor * scale_factor;
    const int blocks = std::min(1024, cdiv(static_cast<long long>(n), WG));   // 1024 partial maxima fit g.pinned
    CHK(g.blockmax.ensure(sizeof(double) * blocks));
    hipLaunchKernelGGL(k_add_clear_kick<true>, dim3(blocks), dim3(WG), 0, s, n, ax, ay, az, bx, by, bz, vx, vy, vz,
                       dt / a3, dt * 2.0 * hubble, g.blockmax.as<double>());
    HIPCHK(hipGetLastError());
    const double* h = static_cast<const double*>(g.pinned);
    HIPCHK(hipMemcpyAsync(g.pinned, g.blockmax.p, sizeof(double) * blocks, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    double m = 1e-9;
    for (int b = 0; b < blocks; ++b) m = (h[b] > m) ? h[b] : m;  // strict '>' as src/integrator.cpp:209
    *out_host = m;
    return NBPP_OK;
}

int nbpp_max_accel_sq_device(size_t n, const double* ax, const double* ay, const double* az,
                             double* out_host, void* stream) {
    if (!g.ready) return fail(NBPP_ERR_NOINIT, "nbpp_init has not been called");
    if (!out_host) return fail(NBPP_ERR_ARG, "out_host is NULL");
    *out_host = 1e-9;  // the floor of src/integrator.cpp:201
    if (n == 0) return NBPP_OK;
    if (!ax || !ay || !az) return fail(NBPP_ERR_ARG, "NULL array pointer");
    std::lock_guard<std::mutex> lock(g_mu);
    HIPCHK(hipSetDevice(g.device));
    hipStream_t s = static_cast<hipStream_t>(stream);   // NULL: the HIP null stream
    const int blocks = std::min(1024, cdiv(static_cast<long long>(n), WG));
    CHK(g.blockmax.ensure(sizeof(double) * blocks));
    hipLaunchKernelGGL(k_max_accel_sq, dim3(blocks), dim3(WG), 0, s, n, ax, ay, az,
                       g.blockmax.as<double>());
    HIPCHK(hipGetLastError());
    const double* h = static_cast<const double*>(g.pinned);   // 1024 doubles fit
    HIPCHK(hipMemcpyAsync(g.pinned, g.blockmax.p, sizeof(double) * blocks, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    double m = 1e-9;
    for (int b = 0; b < blocks; ++b) m = (h[b] > m) ? h[b] : m;  // strict '>' as src/integrator.cpp:209
    *out_host = m;
    return NBPP_OK;
}

}  // extern "C"
