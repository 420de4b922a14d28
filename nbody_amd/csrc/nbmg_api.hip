// nbmg_api.hip - the C ABI of libnbmg (include/nbmg.h): one process per GPU, particles cut into
// index blocks, positions all-gathered with RCCL per force evaluation, libnbpp for the arithmetic.
// Host code only: the kernels are libnbpp's.
#include <hip/hip_runtime.h>
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
    int64_t opt_timing = 0, opt_pair_shares = 0;
    nbmg_stats stats{};
};

Ctx g;
std::mutex g_mu;

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
    std::lock_guard<std::mutex> lock(g_mu);
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

int nbmg_init_from_env(void) {
    const char* ws = std::getenv("WORLD_SIZE");
    if (!ws || std::atoi(ws) <= 1) return nbmg_init(0, 1, nullptr, 0);
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
    return rc;
}

void nbmg_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g.ready) return;
    (void)hipSetDevice(g.device);
    (void)hipDeviceSynchronize();
    if (g.comm) (void)ncclCommDestroy(g.comm);
    g.comm = nullptr;
    for (Buf& b : g.full) b.release();
    for (Buf& b : g.sums) b.release();
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

int nbmg_accumulate_device(const nbpp_params* p, size_t n, const double* own_x, const double* own_y,
                           const double* own_z, const double* m_full, double* ax, double* ay, double* az,
                           void* stream) {
    if (!g.ready) return fail(NBMG_ERR_NOINIT, "nbmg_init has not been called");
    if (!p) return fail(NBMG_ERR_ARG, "params is NULL");
    if (n == 0) return NBMG_OK;
    std::lock_guard<std::mutex> lock(g_mu);
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
    for (Buf& b : g.full) CHK(b.ensure(padded));
    const bool timing = g.opt_timing != 0;

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
    std::lock_guard<std::mutex> lock(g_mu);
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
