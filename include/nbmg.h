/*
 * nbmg.h - C ABI of the multi-GPU layer over libnbpp: one process per GPU, RCCL over xGMI.
 *
 * The reference is one process with one device boundary, the map(to/tofrom) clauses of its
 * `omp target` region (src/particles.cpp:169-173); it has no multi-process code.  libnbmg.so is
 * the counterpart of that boundary for the GPUs of one node: the particles are cut into
 * contiguous index blocks, one per rank; every force evaluation all-gathers the positions
 * (ncclAllGather, masses once) and each rank then runs libnbpp for its own block of targets
 * against all sources.  Nothing here computes a force: the arithmetic is libnbpp's.
 *
 * Layout.  With stride = ceil(n / nranks), rank r owns the particles
 * [r * stride, min(n, (r + 1) * stride)) (nbmg_partition).  Arrays that hold "all particles" on a
 * device are PADDED arrays of nbmg_padded_length(n) = stride * nranks doubles in global index
 * order, so that rank r's block starts at r * stride and one in-place collective moves whole
 * blocks; entries beyond n are never read as particles.
 *
 * Every function returns NBMG_OK (0) or a negative nbmg_status; nothing throws;
 * nbmg_last_error() returns the message of the last failure on the calling thread.  All ranks
 * must make the collective calls (everything below nbmg_partition) in the same order.
 */
#ifndef NBMG_H
#define NBMG_H

#include <stddef.h>
#include <stdint.h>

#include "nbpp.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum nbmg_status {
    NBMG_OK = 0,
    NBMG_ERR_ARG = -1,
    NBMG_ERR_HIP = -2,
    NBMG_ERR_NOINIT = -3,
    NBMG_ERR_NBPP = -4,        /* a libnbpp call failed (message carries nbpp_last_error())   */
    NBMG_ERR_RCCL = -5,        /* a collective failed: RCCL's message, or on the host transport
                                  "ranks disagree on the collective" / a peer that never came */
    NBMG_ERR_RENDEZVOUS = -6   /* the ranks could not exchange the communicator id            */
} nbmg_status;

#define NBMG_UNIQUE_ID_BYTES 128   /* sizeof(ncclUniqueId) */

/* Device time of the phases of the most recent nbmg_accumulate_device (NBMG_OPT_TIMING on). */
typedef struct nbmg_stats {
    double allgather_ms;      /* the three position all-gathers                               */
    double compute_ms;        /* libnbpp: binning + pair kernel                               */
    double reduce_ms;         /* pair-share mode: reduce-scatter of the three sums + the add  */
    double total_ms;
    uint64_t allgather_bytes; /* bytes this rank received in the all-gathers                  */
    uint64_t reduce_bytes;    /* bytes this rank sent into the reduce-scatters                */
    int32_t pair_shares;      /* 1: every unordered pair once across the ranks + reduce-scatter;
                                 0: this rank's targets against all sources, no reduction     */
    int32_t halo_exchange;    /* 1: the call traded halos instead of all-gathering positions  */
    uint64_t halo_particles;  /* halo exchange: particles this rank received from its peers
                                 (allgather_bytes then counts their 32 bytes each)           */
} nbmg_stats;

enum {
    NBMG_OPT_TIMING = 1,      /* 1 = record HIP-event times into nbmg_stats (waits for the GPU) */
    NBMG_OPT_PAIR_SHARES = 2, /* direct summation: 0 = automatic (shares of the unordered pairs on
                                 more than one rank), 1 = always, -1 = never (i-partition,
                                 all-gather only, bitwise reproducible with
                                 NBPP_OPT_DETERMINISTIC)                                        */
    NBMG_OPT_HALO_EXCHANGE = 3 /* P3M short range on more than one rank: instead of all-gathering
                                 every position, each rank sends a peer only its particles
                                 within a cutoff radius of that peer's block
                                 (nbmg_x_extent_device, nbmg_halo_select_device, ncclSend /
                                 ncclRecv).  0 = automatic (default): halos whenever every
                                 block is a slab in x (its interval, widened by the cutoff
                                 radius on both sides, under 3/4 of the box - every rank      
                                 sees the same intervals and decides alike), else the
                                 all-gather; 1 = always; -1 = never; 2 = that code path on one
                                 rank too (nothing travels; for tests)                          */
};

/*
 * Lifecycle.  Rank 0 obtains a communicator id (nbmg_get_unique_id) and hands its
 * NBMG_UNIQUE_ID_BYTES bytes to the other ranks by any means the launcher has; every rank then
 * calls nbmg_init(rank, nranks, id, device), which binds the process to HIP device `device`
 * (nbpp_init) and creates the RCCL communicator; a failed nbmg_init leaves nothing allocated.
 * nranks == 1 needs no id (id may be NULL).
 * nbmg_init_from_env does both for processes started by a launcher that sets RANK, WORLD_SIZE
 * and LOCAL_RANK (torch.distributed.run, mpirun wrappers): rank 0 writes the id to the file named
 * by NBMG_ID_FILE (default: /tmp/nbmg-<uid>/id_<MASTER_PORT>_<launcher pid or NBMG_RUN_TOKEN>, in a
 * directory only this user can write; an older file of that name is removed first), the others wait
 * for it; without WORLD_SIZE it is nbmg_init(0, 1, NULL, 0).  NBMG_HALO_EXCHANGE in the environment
 * sets the option of that name.
 *
 * Transport.  RCCL, one GPU per rank, is the product.  With NBMG_TRANSPORT=host in rank 0's
 * environment when it draws the id, the id names a POSIX shared-memory segment instead and every
 * rank that is given it carries the same collectives through host memory (device -> segment ->
 * device, a barrier in between).  That is for ranks that SHARE a GPU, which RCCL refuses: a
 * rehearsal of the multi-rank flow on a one-GPU machine, never a measurement.  Its barrier also
 * checks that all ranks are in the same collective and gives NBMG_ERR_RCCL instead of a hang when
 * they are not, or when a peer does not arrive within NBMG_HOST_TRANSPORT_TIMEOUT (120) seconds.
 * NBMG_ONE_DEVICE=<d> makes nbmg_init_from_env put every rank on device d.  nbmg_transport()
 * names the transport in use ("rccl", "host", "" before nbmg_init).
 * Replaces: the implicit single device of the reference's `omp target` region
 * (src/particles.cpp:169).
 */
int nbmg_get_unique_id(void* id);
int nbmg_init(int rank, int nranks, const void* id, int device);
int nbmg_init_from_env(void);
void nbmg_shutdown(void);
const char* nbmg_last_error(void);
int nbmg_rank(void);     /* -1 before nbmg_init */
int nbmg_nranks(void);   /*  0 before nbmg_init */
const char* nbmg_transport(void);
int nbmg_set_option(int key, int64_t value);
int nbmg_get_stats(nbmg_stats* out);

/* The block of rank `rank`: pure host arithmetic, usable without a GPU or a communicator. */
int nbmg_partition(size_t n, int rank, int nranks, size_t* begin, size_t* count);
size_t nbmg_padded_length(size_t n, int nranks);

/*
 * own[0, count) of every rank -> full[0, padded length), global order, on every rank
 * (DEVICE pointers; `full` holds nbmg_padded_length(n) doubles).  Enqueued on `stream`.
 * Replaces: map(to: ...) of an array (src/particles.cpp:169-171) for the ranks that do not own it.
 */
int nbmg_allgather_device(size_t n, const double* own, double* full, void* stream);

/* In-place sum / maximum over the ranks of `count` doubles at DEVICE pointer `data`. */
int nbmg_allreduce_sum_device(size_t count, double* data, void* stream);
int nbmg_allreduce_max_device(size_t count, double* data, void* stream);

/*
 * The multi-GPU compute_pp_forces (src/particles.cpp:133-367):
 *     a[i] += sum_{j != i} a_ij      for the caller's own particles i, j over ALL n particles.
 * own_x/y/z: DEVICE arrays of this rank's `count` positions; m_full: DEVICE array of all n masses
 * (padded length; gather it once with nbmg_allgather_device); ax/ay/az: DEVICE arrays of `count`
 * accelerations, accumulated into.  The call all-gathers the positions into arrays the library
 * owns, then
 *   - P3M (use_pm), or direct summation without pair shares: nbpp_accumulate_device with the
 *     rank's block as the target range; no reduction; bitwise reproducible with
 *     NBPP_OPT_DETERMINISTIC;
 *   - direct summation with pair shares (default on more than one rank):
 *     nbpp_accumulate_pairs_device evaluates share `rank` of the unordered pairs into zeroed
 *     full-length sums, ncclReduceScatter leaves each rank the complete sums of its block, which
 *     are added to ax/ay/az.  1.6x the pair rate of the ordered kernel for 3 x 8 bytes per
 *     particle of extra traffic.
 * Everything is enqueued on `stream` in order.
 */
int nbmg_accumulate_device(const nbpp_params* p, size_t n,
                           const double* own_x, const double* own_y, const double* own_z,
                           const double* m_full, double* ax, double* ay, double* az, void* stream);

/*
 * The pieces of the halo exchange (NBMG_OPT_HALO_EXCHANGE), usable on their own; both block until
 * their result is on the host.
 * nbmg_x_extent_device: the circular x interval [*lo, *hi] that holds all `count` positions x
 *     (DEVICE pointer): the complement of the longest empty stretch of the box, to 1/4096 of the
 *     box; *lo > *hi means the interval wraps through x = 0; [0, L] when there is no gap.
 * nbmg_halo_select_device: the particles whose x lies within `margin` of the circular interval
 *     [lo, hi], in index order, packed into `out` (DEVICE, 4 * capacity doubles: x at out[0..),
 *     y at out[capacity..), z at out[2 capacity..), m at out[3 capacity..)); *n_selected is
 *     their number.  out == NULL only counts.
 * No counterpart in the reference (one process, one device boundary, src/particles.cpp:169-173).
 */
int nbmg_x_extent_device(size_t count, const double* x, double domain_size, double* lo, double* hi,
                         void* stream);
int nbmg_halo_select_device(size_t count, const double* x, const double* y, const double* z,
                            const double* m, double domain_size, double lo, double hi, double margin,
                            double* out, size_t capacity, size_t* n_selected, void* stream);

/*
 * max(1e-9, max over ALL ranks' particles of ax^2 + ay^2 + az^2) -> *out_host on every rank
 * (blocking): calculate_max_acceleration (src/integrator.cpp:200-213) across the partition, so
 * that every rank derives the same gravity time step.
 */
int nbmg_max_accel_sq_device(size_t count, const double* ax, const double* ay, const double* az,
                             double* out_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NBMG_H */
