// nbmg_api.hip - the C ABI of libnbmg (include/nbmg.h): one process per GPU, particles cut into
// index blocks, positions all-gathered with RCCL per force evaluation (or, for the P3M short range,
// only the particles within a cutoff radius of a peer's block sent to that peer), libnbpp for the
// arithmetic.  The only kernels here select and pack those halo particles.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
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
        // the fill runs on the null stream, which a caller's non-blocking stream does not wait for:
        // without this wait it could land on top of the first data that stream copies in
        HIPCHK(hipDeviceSynchronize());
        cap = count;
        return NBMG_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// ---- how the ranks exchange data ---------------------------------------------------------------
// The collectives the library issues, at the granularity it issues them (all of them enqueued on,
// or ordered with, the caller's stream; all in place on padded arrays of whole blocks).
struct Xfer {
    double* p;
    size_t count;
    int peer;
};

struct Transport {
    virtual ~Transport() {}
    virtual const char* name() const = 0;
    // arrays[k] + rank * stride holds this rank's block; afterwards every block of every array
    virtual int allgather(double* const* arrays, int n_arrays, size_t stride, hipStream_t s) = 0;
    // arrays[k] holds nranks blocks of partial sums; afterwards arrays[k] + rank * stride holds the
    // sum over the ranks of that block
    virtual int reduce_scatter_sum(double* const* arrays, int n_arrays, size_t stride, hipStream_t s) = 0;
    virtual int allreduce(double* data, size_t count, bool maximum, hipStream_t s) = 0;
    // point-to-point: message k of the sends to peer q is received by q as message k of its
    // receives from this rank
    virtual int exchange(const std::vector<Xfer>& sends, const std::vector<Xfer>& recvs, hipStream_t s) = 0;
};

// RCCL over xGMI: one GPU per rank.  This is the product.
struct RcclTransport final : Transport {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    ~RcclTransport() override {
        if (comm) (void)ncclCommDestroy(comm);
    }
    int start(int rank_, int nranks_, const void* id) {
        rank = rank_;
        nranks = nranks_;
        ncclUniqueId u;
        if (id) std::memcpy(&u, id, sizeof(u));
        else NCCLCHK(ncclGetUniqueId(&u));   // a communicator of one
        NCCLCHK(ncclCommInitRank(&comm, nranks, u, rank));
        return NBMG_OK;
    }
    const char* name() const override { return "rccl"; }
    int allgather(double* const* arrays, int n_arrays, size_t stride, hipStream_t s) override {
        NCCLCHK(ncclGroupStart());
        for (int k = 0; k < n_arrays; ++k)
            NCCLCHK(ncclAllGather(arrays[k] + static_cast<size_t>(rank) * stride, arrays[k], stride, ncclDouble, comm, s));
        NCCLCHK(ncclGroupEnd());
        return NBMG_OK;
    }
    int reduce_scatter_sum(double* const* arrays, int n_arrays, size_t stride, hipStream_t s) override {
        NCCLCHK(ncclGroupStart());
        for (int k = 0; k < n_arrays; ++k)
            NCCLCHK(ncclReduceScatter(arrays[k], arrays[k] + static_cast<size_t>(rank) * stride, stride, ncclDouble,
                                      ncclSum, comm, s));
        NCCLCHK(ncclGroupEnd());
        return NBMG_OK;
    }
    int allreduce(double* data, size_t count, bool maximum, hipStream_t s) override {
        NCCLCHK(ncclAllReduce(data, data, count, ncclDouble, maximum ? ncclMax : ncclSum, comm, s));
        return NBMG_OK;
    }
    int exchange(const std::vector<Xfer>& sends, const std::vector<Xfer>& recvs, hipStream_t s) override {
        NCCLCHK(ncclGroupStart());
        for (const Xfer& x : sends) NCCLCHK(ncclSend(x.p, x.count, ncclDouble, x.peer, comm, s));
        for (const Xfer& x : recvs) NCCLCHK(ncclRecv(x.p, x.count, ncclDouble, x.peer, comm, s));
        NCCLCHK(ncclGroupEnd());
        return NBMG_OK;
    }
};

// Host-staged collectives through one POSIX shared-memory segment, for ranks that SHARE a device
// (RCCL refuses two ranks on one GPU): every collective waits for the stream, copies device ->
// segment, meets the other ranks at a barrier and copies segment -> device.  Slow by design; it
// exists so that the multi-rank flow of this library - partition, padded collectives, pair shares,
// halo trade, the engine's collective order - runs for real on a machine with one GPU.  Never chosen
// by itself: rank 0 asks for it (NBMG_TRANSPORT=host when it draws the communicator id) and the id
// carries the choice to the others.  Every collective posts what it is about to do; ranks that
// disagree (one in an all-gather, one in an all-reduce: the classic cause of a hang) get an error
// from the barrier instead of waiting for ever, and so does a rank whose peers never arrive.
constexpr char HOST_ID_MAGIC[8] = {'N', 'B', 'M', 'G', 'H', 'O', 'S', 'T'};

struct HostTransport final : Transport {
    struct Header {
        std::atomic<uint32_t> ready;        // set by rank 0 once the header is initialised
        std::atomic<uint32_t> arrived;      // ranks waiting at the current barrier
        std::atomic<uint32_t> generation;   // barriers completed
        std::atomic<uint32_t> attached;     // ranks that mapped the segment
        std::atomic<uint32_t> failed;       // a barrier found the ranks in different collectives
        uint64_t capacity;                  // doubles in the data area
        uint64_t posted[64][2];             // per rank: what its current collective is (code, size)
    };
    static constexpr int MAX_RANKS = 64;
    int rank = 0, nranks = 1;
    std::string shm_name;
    Header* head = nullptr;
    double* data = nullptr;
    size_t mapped = 0;
    int timeout_s = 120;

    ~HostTransport() override {
        if (head) {
            (void)::munmap(head, mapped);
            if (rank == 0) (void)::shm_unlink(shm_name.c_str());
        }
    }
    const char* name() const override { return "host"; }

    int start(int rank_, int nranks_, const unsigned char* id) {
        rank = rank_;
        nranks = nranks_;
        if (nranks > MAX_RANKS) return fail(NBMG_ERR_ARG, "the host transport takes at most %d ranks", MAX_RANKS);
        char token[33];
        for (int i = 0; i < 16; ++i) std::snprintf(token + 2 * i, 3, "%02x", id[8 + i]);
        shm_name = std::string("/nbmg_") + token;
        size_t bytes = 64ull << 20;
        if (const char* e = std::getenv("NBMG_HOST_TRANSPORT_BYTES")) bytes = std::max<size_t>(1 << 16, std::strtoull(e, nullptr, 10));
        if (const char* e = std::getenv("NBMG_HOST_TRANSPORT_TIMEOUT")) timeout_s = std::max(1, std::atoi(e));
        mapped = sizeof(Header) + bytes;
        const auto give_up = std::chrono::steady_clock::now() + std::chrono::seconds(timeout_s);
        int fd = -1;
        if (rank == 0) {
            (void)::shm_unlink(shm_name.c_str());
            fd = ::shm_open(shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0) return fail(NBMG_ERR_RENDEZVOUS, "shm_open(%s) failed", shm_name.c_str());
            if (::ftruncate(fd, static_cast<off_t>(mapped)) != 0) {
                ::close(fd);
                return fail(NBMG_ERR_RENDEZVOUS, "cannot size %s to %zu bytes", shm_name.c_str(), mapped);
            }
        } else {
            while (fd < 0) {
                fd = ::shm_open(shm_name.c_str(), O_RDWR, 0600);
                struct stat st;
                if (fd >= 0 && (::fstat(fd, &st) != 0 || static_cast<size_t>(st.st_size) < sizeof(Header))) {
                    ::close(fd);   // rank 0 has not sized it yet
                    fd = -1;
                }
                if (fd < 0) {
                    if (std::chrono::steady_clock::now() > give_up)
                        return fail(NBMG_ERR_RENDEZVOUS, "rank %d: segment %s never appeared", rank, shm_name.c_str());
                    std::this_thread::sleep_for(std::chrono::milliseconds(5));
                }
            }
            struct stat st;
            (void)::fstat(fd, &st);
            mapped = static_cast<size_t>(st.st_size);   // rank 0's size rules
        }
        void* m = ::mmap(nullptr, mapped, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) return fail(NBMG_ERR_RENDEZVOUS, "mmap of %s failed", shm_name.c_str());
        head = static_cast<Header*>(m);
        data = reinterpret_cast<double*>(static_cast<char*>(m) + sizeof(Header));
        if (rank == 0) {
            head->arrived.store(0);
            head->generation.store(0);
            head->attached.store(0);
            head->failed.store(0);
            head->capacity = (mapped - sizeof(Header)) / sizeof(double);
            head->ready.store(0x6e626d67u, std::memory_order_release);
        } else {
            while (head->ready.load(std::memory_order_acquire) != 0x6e626d67u) {
                if (std::chrono::steady_clock::now() > give_up)
                    return fail(NBMG_ERR_RENDEZVOUS, "rank %d: segment %s was never initialised", rank, shm_name.c_str());
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
        }
        head->attached.fetch_add(1);
        return barrier(0, 0);      // returns once every rank has joined, like ncclCommInitRank
    }

    // Every rank posts (code, size) and waits for the others; all must have posted the same.
    int barrier(uint64_t code, uint64_t size) {
        head->posted[rank][0] = code;
        head->posted[rank][1] = size;
        const uint32_t gen = head->generation.load(std::memory_order_acquire);
        if (head->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == static_cast<uint32_t>(nranks)) {
            // last to arrive: check that everybody is in the same collective, then release
            bool same = true;
            for (int r = 0; r < nranks; ++r)
                same = same && head->posted[r][0] == code && head->posted[r][1] == size;
            head->arrived.store(0, std::memory_order_relaxed);
            if (!same) {
                head->failed.store(1, std::memory_order_relaxed);   // every rank reads it after the release
                head->generation.store(gen + 1, std::memory_order_release);
                return fail(NBMG_ERR_RCCL, "ranks disagree on the collective being issued (rank %d: code %llu, size %llu)",
                            rank, static_cast<unsigned long long>(code), static_cast<unsigned long long>(size));
            }
            head->generation.store(gen + 1, std::memory_order_release);
            return NBMG_OK;
        }
        const auto give_up = std::chrono::steady_clock::now() + std::chrono::seconds(timeout_s);
        int spins = 0;
        while (head->generation.load(std::memory_order_acquire) == gen) {
            if (++spins > 2000) {
                std::this_thread::sleep_for(std::chrono::microseconds(50));
                if (std::chrono::steady_clock::now() > give_up)
                    return fail(NBMG_ERR_RCCL, "rank %d waited %d s at a collective (code %llu) the other ranks never joined",
                                rank, timeout_s, static_cast<unsigned long long>(code));
            }
        }
        if (head->failed.load(std::memory_order_relaxed))
            return fail(NBMG_ERR_RCCL, "ranks disagree on the collective being issued (rank %d: code %llu, size %llu)", rank,
                        static_cast<unsigned long long>(code), static_cast<unsigned long long>(size));
        return NBMG_OK;
    }

    size_t chunk() const { return std::max<size_t>(1, head->capacity / static_cast<size_t>(nranks)); }

    int allgather(double* const* arrays, int n_arrays, size_t stride, hipStream_t s) override {
        HIPCHK(hipStreamSynchronize(s));
        const size_t C = chunk();
        for (int k = 0; k < n_arrays; ++k)
            for (size_t o = 0; o < stride; o += C) {
                const size_t c = std::min(C, stride - o);
                HIPCHK(hipMemcpy(data + rank * C, arrays[k] + rank * stride + o, sizeof(double) * c, hipMemcpyDeviceToHost));
                CHK(barrier(1, stride));
                for (int r = 0; r < nranks; ++r)
                    if (r != rank)
                        HIPCHK(hipMemcpy(arrays[k] + r * stride + o, data + r * C, sizeof(double) * c, hipMemcpyHostToDevice));
                CHK(barrier(1, stride));
            }
        return NBMG_OK;
    }

    int reduce_scatter_sum(double* const* arrays, int n_arrays, size_t stride, hipStream_t s) override {
        HIPCHK(hipStreamSynchronize(s));
        const size_t C = chunk();
        std::vector<double> sum;
        for (int k = 0; k < n_arrays; ++k)
            for (int d = 0; d < nranks; ++d)                 // the block that rank d keeps
                for (size_t o = 0; o < stride; o += C) {
                    const size_t c = std::min(C, stride - o);
                    HIPCHK(hipMemcpy(data + rank * C, arrays[k] + d * stride + o, sizeof(double) * c, hipMemcpyDeviceToHost));
                    CHK(barrier(2, stride));
                    if (d == rank) {
                        sum.assign(data, data + c);          // rank order: 0, 1, 2, ...
                        for (int r = 1; r < nranks; ++r)
                            for (size_t i = 0; i < c; ++i) sum[i] += data[r * C + i];
                        HIPCHK(hipMemcpy(arrays[k] + d * stride + o, sum.data(), sizeof(double) * c, hipMemcpyHostToDevice));
                    }
                    CHK(barrier(2, stride));
                }
        return NBMG_OK;
    }

    int allreduce(double* dev, size_t count, bool maximum, hipStream_t s) override {
        HIPCHK(hipStreamSynchronize(s));
        const size_t C = chunk();
        std::vector<double> acc;
        for (size_t o = 0; o < count; o += C) {
            const size_t c = std::min(C, count - o);
            HIPCHK(hipMemcpy(data + rank * C, dev + o, sizeof(double) * c, hipMemcpyDeviceToHost));
            CHK(barrier(maximum ? 4 : 3, count));
            acc.assign(data, data + c);
            for (int r = 1; r < nranks; ++r)
                for (size_t i = 0; i < c; ++i)
                    acc[i] = maximum ? std::max(acc[i], data[r * C + i]) : acc[i] + data[r * C + i];
            HIPCHK(hipMemcpy(dev + o, acc.data(), sizeof(double) * c, hipMemcpyHostToDevice));
            CHK(barrier(maximum ? 4 : 3, count));
        }
        return NBMG_OK;
    }

    // One round per sending rank: it lays its messages out in the segment behind a directory
    // (peer, count per message), everybody else picks out what is addressed to it, in order.
    int exchange(const std::vector<Xfer>& sends, const std::vector<Xfer>& recvs, hipStream_t s) override {
        HIPCHK(hipStreamSynchronize(s));
        for (int src = 0; src < nranks; ++src) {
            if (src == rank) {
                size_t need = 1 + 2 * sends.size();
                for (const Xfer& x : sends) need += x.count;
                if (need > head->capacity)
                    return fail(NBMG_ERR_ARG, "host transport: %zu doubles to send, segment holds %llu (NBMG_HOST_TRANSPORT_BYTES)",
                                need, static_cast<unsigned long long>(head->capacity));
                data[0] = static_cast<double>(sends.size());
                size_t at = 1 + 2 * sends.size();
                for (size_t k = 0; k < sends.size(); ++k) {
                    data[1 + 2 * k] = static_cast<double>(sends[k].peer);
                    data[2 + 2 * k] = static_cast<double>(sends[k].count);
                    if (sends[k].count)
                        HIPCHK(hipMemcpy(data + at, sends[k].p, sizeof(double) * sends[k].count, hipMemcpyDeviceToHost));
                    at += sends[k].count;
                }
            }
            CHK(barrier(5, 0));
            if (src != rank) {
                const size_t n_msgs = static_cast<size_t>(data[0]);
                size_t at = 1 + 2 * n_msgs, mine = 0;
                for (size_t k = 0; k < n_msgs; ++k) {
                    const int peer = static_cast<int>(data[1 + 2 * k]);
                    const size_t count = static_cast<size_t>(data[2 + 2 * k]);
                    if (peer == rank) {
                        // the next of my receives from src
                        while (mine < recvs.size() && recvs[mine].peer != src) ++mine;
                        if (mine == recvs.size() || recvs[mine].count != count)
                            return fail(NBMG_ERR_RCCL, "rank %d: message %zu from rank %d (%zu doubles) has no matching receive",
                                        rank, k, src, count);
                        if (count) HIPCHK(hipMemcpy(recvs[mine].p, data + at, sizeof(double) * count, hipMemcpyHostToDevice));
                        ++mine;
                    }
                    at += count;
                }
                while (mine < recvs.size() && recvs[mine].peer != src) ++mine;
                if (mine != recvs.size())
                    return fail(NBMG_ERR_RCCL, "rank %d: a receive from rank %d was never sent", rank, src);
            }
            CHK(barrier(5, 0));
        }
        return NBMG_OK;
    }
};

struct Ctx {
    bool ready = false;
    int rank = -1, nranks = 0, device = -1;
    Transport* link = nullptr;   // RCCL, or the host-staged rehearsal transport
    Buf full[3];                // all positions, padded, global order
    Buf sums[3];                // pair-share mode: full-length partial sums
    double* scalar = nullptr;   // one device double for small reductions
    double* pinned = nullptr;   // its page-locked host twin
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t opt_timing = 0, opt_pair_shares = 0, opt_halo = 0;
    int halo_rest_calls = 0;    // automatic halo choice: calls left before the blocks are looked at again
    nbmg_stats stats{};
};

Ctx g;
std::recursive_mutex g_mu;   // recursive: the halo exchange calls the public selection entry points

// ---- halo selection: which of my particles lie within `margin` of a peer's x interval ----------
constexpr int XBINS = 4096;      // resolution of a block's x extent

// Positions outside [0, L) are legal input (libnbpp folds them into the box), so the bin is that of
// the periodic image: x = -0.3 L marks the bin of 0.7 L.
__global__ void k_mark_x_bins(size_t n, const double* __restrict__ x, double inv_L, int* __restrict__ mark) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        double t = x[i] * inv_L;
        t -= floor(t);                                   // [0, 1]; 1 only by rounding of a tiny negative
        int b = static_cast<int>(t * XBINS);
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
    const char* want = std::getenv("NBMG_TRANSPORT");
    if (want && std::strcmp(want, "host") == 0) {
        // the id of a host-staged communicator: a magic word and 16 random bytes that name the segment
        unsigned char* out = static_cast<unsigned char*>(id);
        std::memset(out, 0, NBMG_UNIQUE_ID_BYTES);
        std::memcpy(out, HOST_ID_MAGIC, sizeof(HOST_ID_MAGIC));
        FILE* f = std::fopen("/dev/urandom", "rb");
        const bool ok = f && std::fread(out + 8, 1, 16, f) == 16;
        if (f) std::fclose(f);
        if (!ok) return fail(NBMG_ERR_RENDEZVOUS, "cannot read /dev/urandom");
        return NBMG_OK;
    }
    if (want && *want && std::strcmp(want, "rccl") != 0)
        return fail(NBMG_ERR_ARG, "NBMG_TRANSPORT is '%s': 'rccl' (default) or 'host'", want);
    ncclUniqueId u;
    NCCLCHK(ncclGetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return NBMG_OK;
}

namespace {
void release_all() {
    delete g.link;
    g.link = nullptr;
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
}

int init_locked(int rank, int nranks, const void* id, int device) {
    PPCHK(nbpp_init(device));            // binds the process to the device, checks it is gfx950
    HIPCHK(hipGetDevice(&device));
    if (id && std::memcmp(id, HOST_ID_MAGIC, sizeof(HOST_ID_MAGIC)) == 0) {
        HostTransport* t = new HostTransport;
        g.link = t;
        CHK(t->start(rank, nranks, static_cast<const unsigned char*>(id)));
    } else {
        RcclTransport* t = new RcclTransport;
        g.link = t;
        CHK(t->start(rank, nranks, id));
    }
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&g.scalar), 256));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&g.pinned), 256, hipHostMallocDefault));
    for (auto& e : g.ev) HIPCHK(hipEventCreate(&e));
    g.rank = rank;
    g.nranks = nranks;
    g.device = device;
    g.stats = nbmg_stats{};
    g.halo_rest_calls = 0;
    g.ready = true;
    return NBMG_OK;
}
}  // namespace

int nbmg_init(int rank, int nranks, const void* id, int device) {
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    if (g.ready) {
        if (rank == g.rank && nranks == g.nranks) return NBMG_OK;
        return fail(NBMG_ERR_ARG, "already initialised as rank %d of %d", g.rank, g.nranks);
    }
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(NBMG_ERR_ARG, "rank %d is not in [0, %d)", rank, nranks);
    if (nranks > 1 && !id) return fail(NBMG_ERR_ARG, "more than one rank needs the communicator id of rank 0");
    const int rc = init_locked(rank, nranks, id, device);
    if (rc != NBMG_OK) release_all();    // nothing half-built stays behind (the message is kept)
    return rc;
}

const char* nbmg_transport(void) { return g.ready && g.link ? g.link->name() : ""; }

namespace {
int options_from_env(int rc) {
    if (rc != NBMG_OK) return rc;
    if (const char* h = std::getenv("NBMG_HALO_EXCHANGE")) g.opt_halo = std::atoi(h);
    return NBMG_OK;
}

// Where rank 0 leaves the communicator id for the others: NBMG_ID_FILE, else a file in a directory
// only this user can write (/tmp/nbmg-<uid>, mode 0700), named after the launcher's port AND the
// launcher's process id - ranks started by one launcher share their parent - so that neither a
// crashed earlier run nor a concurrent job on the same port can be mistaken for this one.
int id_file_path(std::string* path) {
    if (const char* f = std::getenv("NBMG_ID_FILE")) {
        *path = f;
        return NBMG_OK;
    }
    const std::string dir = "/tmp/nbmg-" + std::to_string(static_cast<unsigned long>(::getuid()));
    if (::mkdir(dir.c_str(), 0700) != 0 && errno != EEXIST) return fail(NBMG_ERR_RENDEZVOUS, "cannot create %s", dir.c_str());
    struct stat st;
    if (::lstat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != ::getuid() || (st.st_mode & 022))
        return fail(NBMG_ERR_RENDEZVOUS, "%s is not a directory that only this user can write", dir.c_str());
    const char* port = std::getenv("MASTER_PORT");
    const char* token = std::getenv("NBMG_RUN_TOKEN");
    *path = dir + "/id_" + (port ? port : "29500") + "_" +
            (token ? std::string(token) : std::to_string(static_cast<long>(::getppid())));
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
    int device = lr ? std::atoi(lr) : rank;
    if (const char* one = std::getenv("NBMG_ONE_DEVICE")) device = std::atoi(one);   // rehearsal: every rank on that device
    std::string path;
    CHK(id_file_path(&path));
    unsigned char id[NBMG_UNIQUE_ID_BYTES];
    if (rank == 0) {
        CHK(nbmg_get_unique_id(id));
        (void)::unlink(path.c_str());                        // whatever an earlier run left behind
        const std::string tmp = path + ".tmp";
        (void)::unlink(tmp.c_str());
        const int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
        if (fd < 0) return fail(NBMG_ERR_RENDEZVOUS, "cannot create %s", tmp.c_str());
        const bool ok = ::write(fd, id, sizeof(id)) == static_cast<ssize_t>(sizeof(id));
        ::close(fd);
        if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0)
            return fail(NBMG_ERR_RENDEZVOUS, "cannot publish %s", path.c_str());
    } else {
        // wait for rank 0's file: ours to read only if this user wrote it, whole, less than a minute ago
        const auto give_up = std::chrono::steady_clock::now() + std::chrono::seconds(120);
        bool have = false;
        while (!have && std::chrono::steady_clock::now() < give_up) {
            const int fd = ::open(path.c_str(), O_RDONLY | O_NOFOLLOW);
            if (fd >= 0) {
                struct stat st;
                if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == ::getuid() &&
                    st.st_size == NBMG_UNIQUE_ID_BYTES && std::time(nullptr) - st.st_mtime < 60)
                    have = ::read(fd, id, sizeof(id)) == static_cast<ssize_t>(sizeof(id));
                ::close(fd);
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
    release_all();
    nbpp_shutdown();
}

int nbmg_set_option(int key, int64_t value) {
    switch (key) {
        case NBMG_OPT_TIMING: g.opt_timing = value; return NBMG_OK;
        case NBMG_OPT_PAIR_SHARES: g.opt_pair_shares = value; return NBMG_OK;
        case NBMG_OPT_HALO_EXCHANGE:
            g.opt_halo = value;
            g.halo_rest_calls = 0;
            return NBMG_OK;
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
    double* arrays[1] = {full};
    return g.link->allgather(arrays, 1, stride, s);
}

int nbmg_allreduce_sum_device(size_t count, double* data, void* stream) {
    if (!g.ready) return fail(NBMG_ERR_NOINIT, "nbmg_init has not been called");
    if (count == 0) return NBMG_OK;
    if (!data) return fail(NBMG_ERR_ARG, "NULL array pointer");
    HIPCHK(hipSetDevice(g.device));
    return g.link->allreduce(data, count, false, static_cast<hipStream_t>(stream));
}

int nbmg_allreduce_max_device(size_t count, double* data, void* stream) {
    if (!g.ready) return fail(NBMG_ERR_NOINIT, "nbmg_init has not been called");
    if (count == 0) return NBMG_OK;
    if (!data) return fail(NBMG_ERR_ARG, "NULL array pointer");
    HIPCHK(hipSetDevice(g.device));
    return g.link->allreduce(data, count, true, static_cast<hipStream_t>(stream));
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
    hipLaunchKernelGGL(k_mark_x_bins, dim3(blocks_for(count)), dim3(256), 0, s, count, x, 1.0 / domain_size, g_bins.p);
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
// `only_slabs`: give up (*used = false, nothing added) unless every block's interval, widened by
// the cutoff radius on both sides, covers less than three quarters of the box (beyond that a rank
// would receive most of the particles anyway) - the same decision on every
// rank, since all of them see the same intervals.
int accumulate_with_halos(const nbpp_params* p, size_t count, const double* own_x, const double* own_y,
                          const double* own_z, const double* own_m, double* ax, double* ay, double* az,
                          bool only_slabs, bool* used, hipStream_t s) {
    const int R = g.nranks;
    const double L = p->domain_size;
    const double margin = std::sqrt(p->cutoff_sq) * (1.0 + 1e-9);
    *used = false;
    // 1. everybody's interval
    double mine[2];
    CHK(nbmg_x_extent_device(count, own_x, L, &mine[0], &mine[1], s));
    if (count == 0) mine[0] = mine[1] = -1.0;                // an empty block needs nothing
    Buf& small = g_small;
    CHK(small.ensure(static_cast<size_t>(2 * R + R * R) + 8));
    std::vector<double> iv(2 * R);
    HIPCHK(hipMemcpyAsync(small.p + 2 * g.rank, mine, sizeof(mine), hipMemcpyHostToDevice, s));
    double* small_arrays[1] = {small.p};
    CHK(g.link->allgather(small_arrays, 1, 2, s));
    HIPCHK(hipMemcpyAsync(iv.data(), small.p, sizeof(double) * 2 * R, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    auto circ = [L](double d) { return d - L * std::floor(d / L); };
    auto width_of = [L](double lo, double hi) { return hi > lo ? hi - lo : hi - lo + L; };
    if (only_slabs)
        for (int r = 0; r < R; ++r)
            if (iv[2 * r] >= 0.0 && width_of(iv[2 * r], iv[2 * r + 1]) + 2.0 * margin >= 0.75 * L) return NBMG_OK;
    *used = true;
    // 2. what I owe every peer: count, then pack
    // (a peer whose interval, widened by the margin, does not meet mine gets nothing)
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
    double* mat_arrays[1] = {mat};
    CHK(g.link->allgather(mat_arrays, 1, static_cast<size_t>(R), s));
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
    std::vector<Xfer> sends, recvs;
    for (int r = 0; r < R; ++r) {
        if (r == g.rank) continue;
        for (int c = 0; c < 4; ++c) {
            if (give[r]) sends.push_back(Xfer{g_send.p + give_off[r] + c * give[r], give[r], r});
            if (take[r]) recvs.push_back(Xfer{g_src[c].p + take_off[r], take[r], r});
        }
    }
    CHK(g.link->exchange(sends, recvs, s));
    g.stats.allgather_bytes = 4ull * sizeof(double) * (n_src - count);
    g.stats.halo_particles = n_src - count;
    g.stats.halo_exchange = 1;
    if (g.opt_timing) HIPCHK(hipEventRecord(g.ev[1], s));
    // 5. the pair force of the own block
    if (count)
        PPCHK(nbpp_accumulate_device(p, n_src, g_src[0].p, g_src[1].p, g_src[2].p, g_src[3].p, 0, count, ax, ay, az, s));
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
    // P3M on several ranks: only the particles near a peer's block travel, as long as the blocks are
    // slabs in x (option 0: looked at per call; after a call that found they are not, the next 64
    // calls go straight to the all-gather).
    const bool halo_forced = (g.nranks > 1 && g.opt_halo == 1) || g.opt_halo > 1;
    const bool halo_auto = g.nranks > 1 && g.opt_halo == 0 && g.halo_rest_calls == 0;
    if (g.halo_rest_calls > 0) --g.halo_rest_calls;
    if (p->use_pm && (halo_forced || halo_auto)) {
        if (timing) HIPCHK(hipEventRecord(g.ev[0], s));
        bool used = false;
        CHK(accumulate_with_halos(p, count, own_x, own_y, own_z, m_full + begin, ax, ay, az, !halo_forced, &used, s));
        if (used) {
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
        g.halo_rest_calls = 64;
    }
    for (Buf& b : g.full) CHK(b.ensure(padded));

    // 1. positions of every rank, in global order
    if (timing) HIPCHK(hipEventRecord(g.ev[0], s));
    const double* own[3] = {own_x, own_y, own_z};
    for (int c = 0; c < 3; ++c)
        if (count)
            HIPCHK(hipMemcpyAsync(g.full[c].p + begin, own[c], sizeof(double) * count, hipMemcpyDeviceToDevice, s));
    double* full[3] = {g.full[0].p, g.full[1].p, g.full[2].p};
    CHK(g.link->allgather(full, 3, stride, s));
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
        double* sums[3] = {g.sums[0].p, g.sums[1].p, g.sums[2].p};
        CHK(g.link->reduce_scatter_sum(sums, 3, stride, s));
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
    CHK(g.link->allreduce(g.scalar, 1, true, s));
    HIPCHK(hipMemcpyAsync(g.pinned, g.scalar, sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *out_host = *g.pinned;
    return NBMG_OK;
}

}  // extern "C"
