// nbmg_api.hip - the C ABI of libnbmg (include/nbmg.h): one process per GPU, particles cut into
// index blocks, positions all-gathered with RCCL per force evaluation (or, for the P3M short range,
// only the particles within a cutoff radius of a peer's block sent to that peer), libnbpp for the
// arithmetic.  The only kernels here select and pack those halo particles.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nbmg.h"

namespace {

thread_local char g_err[640] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(NBMG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                  \
    } while (0)

#define NCCLCHK(expr)                                                                         \
    do {                                                                                      \
        ncclResult_t r_ = (expr);                                                             \
        if (r_ != ncclSuccess)                                                                \
            return fail(NBMG_ERR_RCCL, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), \
                        __FILE__, __LINE__);                                                  \
    } while (0)

#define PPCHK(expr)                                                                           \
    do {                                                                                      \
        int rc_ = (expr);                                                                     \
        if (rc_ != NBPP_OK) return fail(NBMG_ERR_NBPP, "%s: %s", #expr, nbpp_last_error());   \
    } while (0)

#define CHK(expr)                       \
    do {                                \
        int rc_ = (expr);               \
        if (rc_ != NBMG_OK) return rc_; \
    } while (0)

// Device buffer that only grows; new memory is zero-filled once so that the padding beyond the
// last particle never holds signalling patterns.
struct Buf {
    double* p = nullptr;
    size_t cap = 0;
    int ensure(size_t count) {
        if (count <= cap) return NBMG_OK;
        if (p) HIPCHK(hipFree(p));
        p = nullptr;
        cap = 0;
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&p), sizeof(double) * count));
        HIPCHK(hipMemset(p, 0, sizeof(double) * count));
        cap = count;
        return NBMG_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct Ctx {
    bool ready = false;
    int rank = -1, nranks = 0, device = -1;
    ncclComm_t comm = nullptr;
    Buf full[3];                // all positions, padded, global order
    Buf sums[3];                // pair-share mode: full-length partial sums
    double* scalar = nullptr;   // one device double for small reductions
    double* pinned = nullptr;   // its page-locked host twin
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t opt_timing = 0, opt_pair_shares = 0, opt_halo = 0;
    nbmg_stats stats{};
};

Ctx g;
std::recursive_mutex g_mu;   // recursive: the halo exchange calls the public selection entry points

// ---- halo selection: which of my particles lie within `margin` of a peer's x interval ----------
constexpr int XBINS = 4096;      // resolution of a block's x extent

__global__ void k_mark_x_bins(size_t n, const double* __restrict__ x, double inv_bin, int* __restrict__ mark) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        int b = static_cast<int>(x[i] * inv_bin);
        b = b < 0 ? 0 : (b >= XBINS ? XBINS - 1 : b);
        mark[b] = 1;
    }
}

// flag[i] = 1 if x[i] lies in the circular interval that starts at `from` and is `width` long
__global__ void k_halo_flags(size_t n, const double* __restrict__ x, double L, double from, double width,
                             int* __restrict__ flag) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        double d = x[i] - from;
        d -= L * floor(d / L);
        flag[i] = d <= width ? 1 : 0;
    }
}

// the flagged particles, in index order, into out[0..n_sel) (x), out[cap..) (y), out[2 cap..) (z), out[3 cap..) (m)
__global__ void k_halo_pack(size_t n, const int* __restrict__ flag, const int* __restrict__ slot,
                            const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
                            const double* __restrict__ m, size_t cap, double* __restrict__ out) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (!flag[i]) continue;
        const size_t s = static_cast<size_t>(slot[i]);
        if (s >= cap) continue;
        out[s] = x[i];
        out[cap + s] = y[i];
        out[2 * cap + s] = z[i];
        out[3 * cap + s] = m[i];
    }
}

struct IntBuf {
    int* p = nullptr;
    size_t cap = 0;
    int ensure(size_t count) {
        if (count <= cap) return NBMG_OK;
        if (p) HIPCHK(hipFree(p));
        p = nullptr;
        cap = 0;
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&p), sizeof(int) * count));
        cap = count;
        return NBMG_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
IntBuf g_flag, g_slot, g_bins;
Buf g_scan_tmp;                 // hipCUB's temporary storage (bytes / 8, rounded up)
Buf g_small;                    // the ranks' intervals and the matrix of halo counts
Buf g_send, g_src[4];           // packed halos for the peers; own + received sources (x y z m)
int blocks_for(size_t n) { return static_cast<int>(std::min<size_t>(4096, (n + 255) / 256)); }

// flags -> positions among the flagged (exclusive sum) and their number
int count_flagged(size_t n, size_t* n_sel, hipStream_t s) {
    CHK(g_slot.ensure(n + 1));
    size_t bytes = 0;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, g_flag.p, g_slot.p, static_cast<int>(n + 1), s));
    CHK(g_scan_tmp.ensure(bytes / sizeof(double) + 1));
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(g_scan_tmp.p, bytes, g_flag.p, g_slot.p, static_cast<int>(n + 1), s));
    int total = 0;
    HIPCHK(hipMemcpyAsync(&total, g_slot.p + n, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *n_sel = static_cast<size_t>(total);
    return NBMG_OK;
}

size_t stride_of(size_t n, int nranks) { return nranks > 0 ? (n + nranks - 1) / nranks : 0; }

}  // namespace

extern "C" {

const char* nbmg_last_error(void) { return g_err; }
int nbmg_rank(void) { return g.ready ? g.rank : -1; }
int nbmg_nranks(void) { return g.ready ? g.nranks : 0; }

int nbmg_partition(size_t n, int rank, int nranks, size_t* begin, size_t* count) {
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(NBMG_ERR_ARG, "rank %d is not in [0, %d)", rank, nranks);
    if (!begin || !count) return fail(NBMG_ERR_ARG, "NULL output pointer");
    const size_t stride = stride_of(n, nranks);
    const size_t b = std::min(n, static_cast<size_t>(rank) * stride);
    *begin = b;
    *count = std::min(stride, n - b);
    return NBMG_OK;
}

size_t nbmg_padded_length(size_t n, int nranks) { return nranks > 0 ? stride_of(n, nranks) * nranks : 0; }

int nbmg_get_unique_id(void* id) {
    if (!id) return fail(NBMG_ERR_ARG, "id is NULL");
    static_assert(sizeof(ncclUniqueId) == NBMG_UNIQUE_ID_BYTES, "NBMG_UNIQUE_ID_BYTES out of date");
    ncclUniqueId u;
    NCCLCHK(ncclGetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return NBMG_OK;
}

int nbmg_init(int rank, int nranks, const void* id, int device) {
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    if (g.ready) {
        if (rank == g.rank && nranks == g.nranks) return NBMG_OK;
        return fail(NBMG_ERR_ARG, "already initialised as rank %d of %d", g.rank, g.nranks);
    }
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(NBMG_ERR_ARG, "rank %d is not in [0, %d)", rank, nranks);
    if (nranks > 1 && !id) return fail(NBMG_ERR_ARG, "more than one rank needs the communicator id of rank 0");
    PPCHK(nbpp_init(device));            // binds the process to the device, checks it is gfx950
    HIPCHK(hipGetDevice(&device));
    ncclUniqueId u;
    if (id) std::memcpy(&u, id, sizeof(u));
    else NCCLCHK(ncclGetUniqueId(&u));   // a communicator of one
    NCCLCHK(ncclCommInitRank(&g.comm, nranks, u, rank));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&g.scalar), 256));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&g.pinned), 256, hipHostMallocDefault));
    for (auto& e : g.ev) HIPCHK(hipEventCreate(&e));
    g.rank = rank;
    g.nranks = nranks;
    g.device = device;
    g.stats = nbmg_stats{};
    g.ready = true;
    return NBMG_OK;
}

namespace {
int options_from_env(int rc) {
    if (rc != NBMG_OK) return rc;
    if (const char* h = std::getenv("NBMG_HALO_EXCHANGE")) g.opt_halo = std::atoi(h);
    return NBMG_OK;
}
}  // namespace

int nbmg_init_from_env(void) {
    const char* ws = std::getenv("WORLD_SIZE");
    if (!ws || std::atoi(ws) <= 1) return options_from_env(nbmg_init(0, 1, nullptr, 0));
    const int nranks = std::atoi(ws);
    const char* rk = std::getenv("RANK");
    if (!rk) return fail(NBMG_ERR_RENDEZVOUS, "WORLD_SIZE is set but RANK is not");
    const int rank = std::atoi(rk);
    const char* lr = std::getenv("LOCAL_RANK");
    const int device = lr ? std::atoi(lr) : rank;
    std::string path;
    if (const char* f = std::getenv("NBMG_ID_FILE")) path = f;
    else {
        const char* port = std::getenv("MASTER_PORT");
        path = std::string("/tmp/nbmg_id_") + (port ? port : "29500");
    }
    unsigned char id[NBMG_UNIQUE_ID_BYTES];
    if (rank == 0) {
        CHK(nbmg_get_unique_id(id));
        const std::string tmp = path + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f) return fail(NBMG_ERR_RENDEZVOUS, "cannot write %s", tmp.c_str());
        const bool ok = std::fwrite(id, 1, sizeof(id), f) == sizeof(id);
        std::fclose(f);
        if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0)
            return fail(NBMG_ERR_RENDEZVOUS, "cannot publish %s", path.c_str());
    } else {
        // wait for a file no older than a minute (one left behind by a run that crashed is older)
        const auto give_up = std::chrono::steady_clock::now() + std::chrono::seconds(120);
        bool have = false;
        while (!have && std::chrono::steady_clock::now() < give_up) {
            struct stat st;
            if (::stat(path.c_str(), &st) == 0 && st.st_size == NBMG_UNIQUE_ID_BYTES &&
                std::time(nullptr) - st.st_mtime < 60) {
                FILE* f = std::fopen(path.c_str(), "rb");
                if (f) {
                    have = std::fread(id, 1, sizeof(id), f) == sizeof(id);
                    std::fclose(f);
                }
            }
            if (!have) std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
        if (!have) return fail(NBMG_ERR_RENDEZVOUS, "rank %d: no communicator id appeared in %s", rank, path.c_str());
    }
    const int rc = nbmg_init(rank, nranks, id, device);   // returns once every rank has joined
    if (rank == 0) (void)::unlink(path.c_str());
    return options_from_env(rc);
}

void nbmg_shutdown(void) {
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    if (!g.ready) return;
    (void)hipSetDevice(g.device);
    (void)hipDeviceSynchronize();
    if (g.comm) (void)ncclCommDestroy(g.comm);
    g.comm = nullptr;
    for (Buf& b : g.full) b.release();
    for (Buf& b : g.sums) b.release();
    g_flag.release(); g_slot.release(); g_bins.release(); g_scan_tmp.release(); g_small.release(); g_send.release();
    for (Buf& b : g_src) b.release();
    if (g.scalar) (void)hipFree(g.scalar);
    if (g.pinned) (void)hipHostFree(g.pinned);
    g.scalar = g.pinned = nullptr;
    for (auto& e : g.ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    g.ready = false;
    g.rank = -1;
    g.nranks = 0;
    nbpp_shutdown();
}

int nbmg_set_option(int key, int64_t value) {
    switch (key) {
        case NBMG_OPT_TIMING: g.opt_timing = value; return NBMG_OK;
        case NBMG_OPT_PAIR_SHARES: g.opt_pair_shares = value; return NBMG_OK;
        case NBMG_OPT_HALO_EXCHANGE: g.opt_halo = value; return NBMG_OK;
        default: return fail(NBMG_ERR_ARG, "unknown option key %d", key);
    }
}

int nbmg_get_stats(nbmg_stats* out) {
    if (!out) return fail(NBMG_ERR_ARG, "out is NULL");
    *out = g.stats;
    return NBMG_OK;
}

int nbmg_allgather_device(size_t n, const double* own, double* full, void* stream) {
    if (!g.ready) return fail(NBMG_ERR_NOINIT, "nbmg_init has not been called");
    if (n == 0) return NBMG_OK;
    if (!own || !full) return fail(NBMG_ERR_ARG, "NULL array pointer");
    HIPCHK(hipSetDevice(g.device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    size_t begin = 0, count = 0;
    CHK(nbmg_partition(n, g.rank, g.nranks, &begin, &count));
    const size_t stride = stride_of(n, g.nranks);
    // own block into its place, then one in-place all-gather of whole blocks
    if (count && own != full + begin)
        HIPCHK(hipMemcpyAsync(full + begin, own, sizeof(double) * count, hipMemcpyDeviceToDevice, s));
    NCCLCHK(ncclAllGather(full + static_cast<size_t>(g.rank) * stride, full, stride, ncclDouble, g.comm, s));
    return NBMG_OK;
}

int nbmg_allreduce_sum_device(size_t count, double* data, void* stream) {
    if (!g.ready) return fail(NBMG_ERR_NOINIT, "nbmg_init has not been called");
    if (count == 0) return NBMG_OK;
    if (!data) return fail(NBMG_ERR_ARG, "NULL array pointer");
    HIPCHK(hipSetDevice(g.device));
    NCCLCHK(ncclAllReduce(data, data, count, ncclDouble, ncclSum, g.comm, static_cast<hipStream_t>(stream)));
    return NBMG_OK;
}

int nbmg_allreduce_max_device(size_t count, double* data, void* stream) {
    if (!g.ready) return fail(NBMG_ERR_NOINIT, "nbmg_init has not been called");
    if (count == 0) return NBMG_OK;
    if (!data) return fail(NBMG_ERR_ARG, "NULL array pointer");
    HIPCHK(hipSetDevice(g.device));
    NCCLCHK(ncclAllReduce(data, data, count, ncclDouble, ncclMax, g.comm, static_cast<hipStream_t>(stream)));
    return NBMG_OK;
}

int nbmg_x_extent_device(size_t count, const double* x, double domain_size, double* lo, double* hi, void* stream) {
    if (!g.ready) return fail(NBMG_ERR_NOINIT, "nbmg_init has not been called");
    if (!lo || !hi) return fail(NBMG_ERR_ARG, "NULL output pointer");
    if (!(domain_size > 0.0)) return fail(NBMG_ERR_ARG, "domain_size must be positive");
    *lo = 0.0;
    *hi = domain_size;
    if (count == 0) return NBMG_OK;
    if (!x) return fail(NBMG_ERR_ARG, "NULL array pointer");
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    HIPCHK(hipSetDevice(g.device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    CHK(g_bins.ensure(XBINS));
    HIPCHK(hipMemsetAsync(g_bins.p, 0, sizeof(int) * XBINS, s));
    hipLaunchKernelGGL(k_mark_x_bins, dim3(blocks_for(count)), dim3(256), 0, s, count, x, XBINS / domain_size, g_bins.p);
    HIPCHK(hipGetLastError());
    std::vector<int> mark(XBINS);
    HIPCHK(hipMemcpyAsync(mark.data(), g_bins.p, sizeof(int) * XBINS, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    // the block is the complement of the longest run of empty bins on the circle
    int best_start = 0, best_len = 0;
    for (int b = 0; b < XBINS; ++b) {
        if (!mark[b]) continue;
        int len = 0;
        while (len < XBINS && !mark[(b + 1 + len) % XBINS]) ++len;
        if (len > best_len) { best_len = len; best_start = (b + 1) % XBINS; }
    }
    if (best_len == 0) return NBMG_OK;                       // occupied all round: the whole box
    const double bin = domain_size / XBINS;
    *lo = ((best_start + best_len) % XBINS) * bin;           // first occupied bin after the gap
    *hi = best_start * bin;                                  // end of the last occupied bin before it
    if (*hi == 0.0) *hi = domain_size;                       // (lo > hi: the interval wraps)
    return NBMG_OK;
}

int nbmg_halo_select_device(size_t count, const double* x, const double* y, const double* z, const double* m,
                            double domain_size, double lo, double hi, double margin, double* out, size_t capacity,
                            size_t* n_selected, void* stream) {
    if (!g.ready) return fail(NBMG_ERR_NOINIT, "nbmg_init has not been called");
    if (!n_selected) return fail(NBMG_ERR_ARG, "n_selected is NULL");
    *n_selected = 0;
    if (count == 0) return NBMG_OK;
    if (!x || !y || !z || !m) return fail(NBMG_ERR_ARG, "NULL array pointer");
    if (!(domain_size > 0.0) || !(margin >= 0.0)) return fail(NBMG_ERR_ARG, "domain_size must be positive, margin >= 0");
    if (count > 0x7ffffff0ull) return fail(NBMG_ERR_ARG, "at most 2^31 - 16 particles per block");
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    HIPCHK(hipSetDevice(g.device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    double width = hi - lo;                                  // circular length of [lo, hi]
    if (width <= 0.0) width += domain_size;
    width += 2.0 * margin;
    double from = lo - margin;
    if (width >= domain_size) { from = 0.0; width = 2.0 * domain_size; }   // everything
    CHK(g_flag.ensure(count + 1));
    HIPCHK(hipMemsetAsync(g_flag.p + count, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_halo_flags, dim3(blocks_for(count)), dim3(256), 0, s, count, x, domain_size, from, width, g_flag.p);
    HIPCHK(hipGetLastError());
    CHK(count_flagged(count, n_selected, s));
    if (!out) return NBMG_OK;                                // counting only
    if (*n_selected > capacity)
        return fail(NBMG_ERR_ARG, "%zu particles selected, room for %zu", *n_selected, capacity);
    hipLaunchKernelGGL(k_halo_pack, dim3(blocks_for(count)), dim3(256), 0, s, count, g_flag.p, g_slot.p, x, y, z, m,
                       capacity, out);
    HIPCHK(hipGetLastError());
    return NBMG_OK;
}

namespace {
// P3M short range with a halo exchange instead of the all-gather: every rank learns the x
// interval of every block, packs for each peer the particles of its own that lie within the
// cutoff radius of that peer's interval, and the peers trade the packs (grouped ncclSend /
// ncclRecv); libnbpp then works on own + received particles with the own block as targets.
int accumulate_with_halos(const nbpp_params* p, size_t n, size_t count, const double* own_x, const double* own_y,
                          const double* own_z, const double* own_m, double* ax, double* ay, double* az,
                          hipStream_t s) {
    const int R = g.nranks;
    const double L = p->domain_size;
    const double margin = std::sqrt(p->cutoff_sq) * (1.0 + 1e-9);
    // 1. everybody's interval
    double mine[2];
    CHK(nbmg_x_extent_device(count, own_x, L, &mine[0], &mine[1], s));
    if (count == 0) mine[0] = mine[1] = -1.0;                // an empty block needs nothing
    Buf& small = g_small;
    CHK(small.ensure(static_cast<size_t>(2 * R + R * R) + 8));
    std::vector<double> iv(2 * R);
    HIPCHK(hipMemcpyAsync(small.p + 2 * g.rank, mine, sizeof(mine), hipMemcpyHostToDevice, s));
    NCCLCHK(ncclAllGather(small.p + 2 * g.rank, small.p, 2, ncclDouble, g.comm, s));
    HIPCHK(hipMemcpyAsync(iv.data(), small.p, sizeof(double) * 2 * R, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    // 2. what I owe every peer: count, then pack
    // (a peer whose interval, widened by the margin, does not meet mine gets nothing)
    auto circ = [L](double d) { return d - L * std::floor(d / L); };
    auto width_of = [L](double lo, double hi) { return hi > lo ? hi - lo : hi - lo + L; };
    const double my_w = width_of(mine[0], mine[1]);
    std::vector<size_t> give(R, 0), give_off(R + 1, 0);
    for (int r = 0; r < R; ++r) {
        if (r == g.rank || count == 0 || iv[2 * r] < 0.0) continue;
        const double peer_w = width_of(iv[2 * r], iv[2 * r + 1]) + 2.0 * margin;
        const double peer_from = iv[2 * r] - margin;
        if (peer_w < L && my_w < L && circ(peer_from - mine[0]) > my_w && circ(mine[0] - peer_from) > peer_w) continue;
        CHK(nbmg_halo_select_device(count, own_x, own_y, own_z, own_m, L, iv[2 * r], iv[2 * r + 1], margin, nullptr, 0,
                                    &give[r], s));
    }
    for (int r = 0; r < R; ++r) give_off[r + 1] = give_off[r] + 4 * give[r];
    CHK(g_send.ensure(std::max<size_t>(give_off[R], 1)));
    for (int r = 0; r < R; ++r) {
        if (r == g.rank || give[r] == 0) continue;
        size_t again = 0;
        CHK(nbmg_halo_select_device(count, own_x, own_y, own_z, own_m, L, iv[2 * r], iv[2 * r + 1], margin,
                                    g_send.p + give_off[r], give[r], &again, s));
    }
    // 3. the count matrix: row r = what rank r gives to each rank
    std::vector<double> row(R), all(static_cast<size_t>(R) * R);
    for (int r = 0; r < R; ++r) row[r] = static_cast<double>(give[r]);
    double* mat = small.p + 2 * R;
    HIPCHK(hipMemcpyAsync(mat + static_cast<size_t>(g.rank) * R, row.data(), sizeof(double) * R, hipMemcpyHostToDevice, s));
    NCCLCHK(ncclAllGather(mat + static_cast<size_t>(g.rank) * R, mat, R, ncclDouble, g.comm, s));
    HIPCHK(hipMemcpyAsync(all.data(), mat, sizeof(double) * R * R, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    std::vector<size_t> take(R, 0), take_off(R + 1, 0);
    for (int r = 0; r < R; ++r) take[r] = r == g.rank ? 0 : static_cast<size_t>(all[static_cast<size_t>(r) * R + g.rank]);
    take_off[0] = count;
    for (int r = 0; r < R; ++r) take_off[r + 1] = take_off[r] + take[r];
    const size_t n_src = take_off[R];
    // 4. own particles first, then the peers' packs
    for (Buf& b : g_src) CHK(b.ensure(std::max<size_t>(n_src, 1)));
    const double* own[4] = {own_x, own_y, own_z, own_m};
    for (int c = 0; c < 4; ++c)
        if (count) HIPCHK(hipMemcpyAsync(g_src[c].p, own[c], sizeof(double) * count, hipMemcpyDeviceToDevice, s));
    NCCLCHK(ncclGroupStart());
    for (int r = 0; r < R; ++r) {
        if (r == g.rank) continue;
        for (int c = 0; c < 4; ++c) {
            if (give[r]) NCCLCHK(ncclSend(g_send.p + give_off[r] + c * give[r], give[r], ncclDouble, r, g.comm, s));
            if (take[r]) NCCLCHK(ncclRecv(g_src[c].p + take_off[r], take[r], ncclDouble, r, g.comm, s));
        }
    }
    NCCLCHK(ncclGroupEnd());
    g.stats.allgather_bytes = 4ull * sizeof(double) * (n_src - count);
    g.stats.halo_particles = n_src - count;
    if (g.opt_timing) HIPCHK(hipEventRecord(g.ev[1], s));
    // 5. the pair force of the own block
    if (count)
        PPCHK(nbpp_accumulate_device(p, n_src, g_src[0].p, g_src[1].p, g_src[2].p, g_src[3].p, 0, count, ax, ay, az, s));
    (void)n;
    return NBMG_OK;
}
}  // namespace

int nbmg_accumulate_device(const nbpp_params* p, size_t n, const double* own_x, const double* own_y,
                           const double* own_z, const double* m_full, double* ax, double* ay, double* az,
                           void* stream) {
    if (!g.ready) return fail(NBMG_ERR_NOINIT, "nbmg_init has not been called");
    if (!p) return fail(NBMG_ERR_ARG, "params is NULL");
    if (n == 0) return NBMG_OK;
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    HIPCHK(hipSetDevice(g.device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    size_t begin = 0, count = 0;
    CHK(nbmg_partition(n, g.rank, g.nranks, &begin, &count));
    if (!m_full || (count && (!own_x || !own_y || !own_z || !ax || !ay || !az)))
        return fail(NBMG_ERR_ARG, "NULL array pointer");
    const size_t stride = stride_of(n, g.nranks);
    const size_t padded = stride * g.nranks;
    const bool pair_shares = !p->use_pm && (g.opt_pair_shares > 0 || (g.opt_pair_shares == 0 && g.nranks > 1));
    g.stats = nbmg_stats{};
    g.stats.pair_shares = pair_shares ? 1 : 0;
    const bool timing = g.opt_timing != 0;
    if (p->use_pm && ((g.nranks > 1 && g.opt_halo > 0) || g.opt_halo > 1)) {
        // only the particles near a peer's block travel
        if (timing) HIPCHK(hipEventRecord(g.ev[0], s));
        CHK(accumulate_with_halos(p, n, count, own_x, own_y, own_z, m_full + begin, ax, ay, az, s));
        if (timing) {
            HIPCHK(hipEventRecord(g.ev[2], s));
            HIPCHK(hipEventSynchronize(g.ev[2]));
            float a = 0, b = 0;
            HIPCHK(hipEventElapsedTime(&a, g.ev[0], g.ev[1]));
            HIPCHK(hipEventElapsedTime(&b, g.ev[1], g.ev[2]));
            g.stats.allgather_ms = a;
            g.stats.compute_ms = b;
            g.stats.total_ms = a + b;
        }
        return NBMG_OK;
    }
    for (Buf& b : g.full) CHK(b.ensure(padded));

    // 1. positions of every rank, in global order
    if (timing) HIPCHK(hipEventRecord(g.ev[0], s));
    const double* own[3] = {own_x, own_y, own_z};
    for (int c = 0; c < 3; ++c)
        if (count)
            HIPCHK(hipMemcpyAsync(g.full[c].p + begin, own[c], sizeof(double) * count, hipMemcpyDeviceToDevice, s));
    NCCLCHK(ncclGroupStart());
    for (int c = 0; c < 3; ++c)
        NCCLCHK(ncclAllGather(g.full[c].p + static_cast<size_t>(g.rank) * stride, g.full[c].p, stride, ncclDouble,
                              g.comm, s));
    NCCLCHK(ncclGroupEnd());
    g.stats.allgather_bytes = 3ull * sizeof(double) * stride * (g.nranks - 1);
    if (timing) HIPCHK(hipEventRecord(g.ev[1], s));

    // 2. the pair force
    if (!pair_shares) {
        if (count)
            PPCHK(nbpp_accumulate_device(p, n, g.full[0].p, g.full[1].p, g.full[2].p, m_full, begin, count, ax, ay,
                                         az, stream));
        if (timing) {
            HIPCHK(hipEventRecord(g.ev[2], s));
            HIPCHK(hipEventRecord(g.ev[3], s));
        }
    } else {
        for (Buf& b : g.sums) CHK(b.ensure(padded));
        for (Buf& b : g.sums) HIPCHK(hipMemsetAsync(b.p, 0, sizeof(double) * padded, s));
        PPCHK(nbpp_accumulate_pairs_device(p, n, g.full[0].p, g.full[1].p, g.full[2].p, m_full, g.rank, g.nranks,
                                           g.sums[0].p, g.sums[1].p, g.sums[2].p, stream));
        if (timing) HIPCHK(hipEventRecord(g.ev[2], s));
        // 3. every rank's share of every particle's sum, added up where the particle lives
        NCCLCHK(ncclGroupStart());
        for (Buf& b : g.sums)
            NCCLCHK(ncclReduceScatter(b.p, b.p + static_cast<size_t>(g.rank) * stride, stride, ncclDouble, ncclSum,
                                      g.comm, s));
        NCCLCHK(ncclGroupEnd());
        g.stats.reduce_bytes = 3ull * sizeof(double) * stride * (g.nranks - 1);
        if (count)
            PPCHK(nbpp_add_device(count, ax, ay, az, g.sums[0].p + begin, g.sums[1].p + begin, g.sums[2].p + begin,
                                  stream));
        if (timing) HIPCHK(hipEventRecord(g.ev[3], s));
    }
    if (timing) {
        HIPCHK(hipEventSynchronize(g.ev[3]));
        float a = 0, b = 0, c = 0;
        HIPCHK(hipEventElapsedTime(&a, g.ev[0], g.ev[1]));
        HIPCHK(hipEventElapsedTime(&b, g.ev[1], g.ev[2]));
        HIPCHK(hipEventElapsedTime(&c, g.ev[2], g.ev[3]));
        g.stats.allgather_ms = a;
        g.stats.compute_ms = b;
        g.stats.reduce_ms = c;
        g.stats.total_ms = a + b + c;
    }
    return NBMG_OK;
}

int nbmg_max_accel_sq_device(size_t count, const double* ax, const double* ay, const double* az, double* out_host,
                             void* stream) {
    if (!g.ready) return fail(NBMG_ERR_NOINIT, "nbmg_init has not been called");
    if (!out_host) return fail(NBMG_ERR_ARG, "out_host is NULL");
    double local = 1e-9;
    PPCHK(nbpp_max_accel_sq_device(count, ax, ay, az, &local, stream));   // floor 1e-9, also for count == 0
    *out_host = local;
    if (g.nranks == 1) return NBMG_OK;
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    HIPCHK(hipSetDevice(g.device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    *g.pinned = local;
    HIPCHK(hipMemcpyAsync(g.scalar, g.pinned, sizeof(double), hipMemcpyHostToDevice, s));
    NCCLCHK(ncclAllReduce(g.scalar, g.scalar, 1, ncclDouble, ncclMax, g.comm, s));
    HIPCHK(hipMemcpyAsync(g.pinned, g.scalar, sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *out_host = *g.pinned;
    return NBMG_OK;
}

}  // extern "C"
