/*
 * nbpp.h - C ABI of the MI355X (gfx950) particle-particle gravity library.
 *
 * One shared library, libnbpp.so, replaces the body of the reference's
 *     void ParticleSystem::compute_pp_forces(const Config&)
 * (reference src/particles.h:62, src/particles.cpp:133-367): direct-summation
 * PP forces and the P3M short-range tapered force, fp64, periodic minimum-image
 * wrap, Plummer softening, "acc += sum" semantics.  All entry points are plain C:
 * pointers, sizes and PODs only.  Every function returns NBPP_OK (0) or a negative
 * nbpp_status; nothing throws; nbpp_last_error() returns the message of the last
 * failure on the calling thread.
 *
 * There is no CPU fallback: every compute entry point fails with NBPP_ERR_HIP when
 * no gfx950 device can be opened.
 */
#ifndef NBPP_H
#define NBPP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum nbpp_status {
    NBPP_OK = 0,
    NBPP_ERR_ARG = -1,     /* null pointer, negative size, inconsistent parameters */
    NBPP_ERR_HIP = -2,     /* a HIP runtime call failed (message has the HIP error) */
    NBPP_ERR_NOINIT = -3,  /* nbpp_init() has not been called (or failed)          */
    NBPP_ERR_RANGE = -4,   /* a size exceeds what the entry point supports         */
    NBPP_ERR_NOMEM = -5    /* device or host allocation failed                     */
} nbpp_status;

/*
 * The scalars the reference hoists out of Config at the top of compute_pp_forces
 * (src/particles.cpp:134-152), one field per hoisted value, plus the two Config
 * fields its cell search uses (MESH_SIZE, CELL_SIZE).
 *
 *   use_pm == 0 : direct summation.  Every ordered pair (i, j != i) contributes once,
 *                 with the minimum-image displacement and Plummer softening soft_sq.
 *                 cutoff_sq, r_switch*, trans_width, pm_soft_sq are ignored
 *                 (src/particles.cpp:326,331,339 are all guarded by use_pm).
 *   use_pm != 0 : P3M short range.  Pairs with r^2 > cutoff_sq contribute nothing;
 *                 r^2 > r_switch_sq is tapered by S = 2x^3 - 3x^2 + 1,
 *                 x = (r - r_switch)/trans_width; the mesh-softened force
 *                 G m d / (r^2 + pm_soft_sq)^(3/2) is subtracted.
 *
 * mesh_size / cell_size are only used to size the library's own search grid; the
 * library never reads the reference's host linked list.
 */
typedef struct nbpp_params {
    double domain_size;  /* Config::DOMAIN_SIZE              src/particles.cpp:143 */
    double G;            /* Config::G                        src/particles.cpp:144 */
    double soft_sq;      /* Config::SOFTENING_SQUARED        src/particles.cpp:145 */
    double pm_soft_sq;   /* (0.5*CELL_SIZE)^2                src/particles.cpp:146-147 */
    double r_switch;     /* Config::R_SWITCH_START           src/particles.cpp:148 */
    double r_switch_sq;  /* Config::R_SWITCH_START_SQ        src/particles.cpp:149 */
    double cutoff_sq;    /* Config::CUTOFF_RADIUS_SQUARED    src/particles.cpp:150 */
    double trans_width;  /* Config::CUTOFF_TRANSITION_WIDTH  src/particles.cpp:151 */
    double cell_size;    /* Config::CELL_SIZE                src/particles.cpp:136,276 */
    int32_t mesh_size;   /* Config::MESH_SIZE                src/particles.cpp:139 */
    int32_t use_pm;      /* Config::USE_PM                   src/particles.cpp:152 */
} nbpp_params;

/* Timing and work counters of the most recent compute call on this thread's device. */
typedef struct nbpp_stats {
    double pair_kernel_ms;   /* HIP-event time of the pair kernel(s) alone            */
    double prepare_ms;       /* sort / binning / packing kernels                      */
    double total_ms;         /* whole call on the device, first to last kernel        */
    double h2d_ms, d2h_ms;   /* host-pointer entry points only, else 0                */
    uint64_t n_sources;      /* j count                                               */
    uint64_t n_targets;      /* i count                                               */
    uint64_t pairs_evaluated;/* ordered pairs whose r^2 was formed (i==j excluded)    */
    uint64_t pairs_accepted; /* P3M: ordered pairs with r^2 <= cutoff_sq (needs
                                NBPP_OPT_TIMING); direct: equal to pairs_evaluated    */
    uint64_t pair_launches;  /* pair-kernel launches in the call                      */
    uint64_t subtiles_total; /* direct kernels, with NBPP_OPT_TIMING: (wavefront, 64-source
                                sub-tile) visits ...                                   */
    uint64_t subtiles_wrapped; /* ... and how many needed the per-pair minimum image   */
    int32_t  kernel_variant; /* which pair kernel ran (NBPP_KERNEL_*)                 */
    int32_t  p3m_pooled;     /* NBPP_KERNEL_P3M_CELLS: 0 if each lane evaluated its own
                                survivors of the cutoff pre-filter, 1 if they were pooled per
                                wavefront (NBPP_OPT_P3M_POOLED_DRAIN), 2 if every unordered
                                pair was evaluated once (NBPP_OPT_P3M_PAIR_ONCE), 3 if the
                                lanes of a target walked its rows together
                                (NBPP_OPT_P3M_ROW_LANES), 4 if they did and the survivors
                                were pooled (NBPP_OPT_P3M_ROWS_POOLED)                       */
    double   cell_occupancy; /* NBPP_KERNEL_P3M_CELLS: mean over particles of the number of
                                particles in the particle's own search cell (1 + the mean
                                per cell when evenly spread; grows with clustering)          */
} nbpp_stats;

enum {
    NBPP_KERNEL_NONE = 0,
    NBPP_KERNEL_DIRECT = 1,       /* all-pairs, tiles classified for the image shift   */
    NBPP_KERNEL_DIRECT_WRAP = 2,  /* all-pairs, per-pair minimum image everywhere      */
    NBPP_KERNEL_P3M_CELLS = 3,    /* counting-sort cell list, cutoff-bounded search    */
    NBPP_KERNEL_P3M_ALLPAIRS = 4, /* P3M pair function over all pairs (small n, tests) */
    NBPP_KERNEL_DIRECT_SYMMETRIC = 5 /* all-pairs, each unordered pair evaluated once and
                                     applied in both directions (fp64 atomics: results are
                                     not bitwise reproducible from run to run)          */
};

/* nbpp_set_option keys */
enum {
    NBPP_OPT_FORCE_KERNEL = 1,   /* 0 = automatic, else one of NBPP_KERNEL_*           */
    NBPP_OPT_TIMING = 2,         /* 1 = record HIP-event timings into nbpp_stats        */
    NBPP_OPT_TARGETS_PER_LANE = 3, /* all-pairs kernels: 0 = automatic, else 1, 2 or 4  */
    NBPP_OPT_SOURCE_SPLIT = 4,   /* all-pairs kernels: 0 = automatic, else the number of
                                    source slices summed separately (blockIdx.y)        */
    NBPP_OPT_P3M_SEARCH_RADIUS = 5, /* P3M: cells per cutoff radius of the search grid,
                                    0 = automatic                                        */
    NBPP_OPT_NO_EQUAL_MASS_PATH = 6, /* 1 = always multiply by the source mass per pair, even
                                    when all masses are equal (testing)                  */
    NBPP_OPT_DETERMINISTIC = 7,  /* 1 = results bitwise reproducible from run to run: never
                                    NBPP_KERNEL_DIRECT_SYMMETRIC nor the pooled or pair-once P3M
                                    kernels by themselves, and the members of every search cell
                                    put into a fixed order before the P3M pair kernel (by default
                                    their order, and so the last bits of the sums, is whatever
                                    the binning's atomics left)                              */
    NBPP_OPT_P3M_LANES_PER_TARGET = 8, /* P3M cell kernel: 0 = automatic (more lanes the fewer
                                    targets), else 1, 2, 4, 8 or 16                       */
    NBPP_OPT_P3M_POOLED_DRAIN = 9, /* P3M cell kernel with one lane per target: 0 = automatic (pool
                                    the pre-filter's survivors per wavefront and evaluate them 64
                                    at a time, sums added with LDS fp64 atomics, unless
                                    NBPP_OPT_DETERMINISTIC is set), 1 = always, -1 = never (every
                                    lane evaluates its own survivors, bitwise reproducible)  */
    NBPP_OPT_P3M_PAIR_ONCE = 10, /* P3M cell kernel, all particles as targets: 0 = automatic
                                    (evaluate every unordered pair once and apply it to both
                                    members, reactions gathered per brick of cells in LDS and
                                    added to the outputs with fp64 atomics, when the particle
                                    set is even enough for a brick's neighbourhood to fit in
                                    LDS and NBPP_OPT_DETERMINISTIC is off; not tried in the call
                                    after one that found the particles clustered,
                                    cell_occupancy >= 6.5), 1 = whenever it fits, -1 = never   */
    NBPP_OPT_P3M_ROW_LANES = 12, /* P3M cell kernel on small sets: lanes of one target that walk every
                                    row of its neighbourhood together, each taking every G-th
                                    candidate (balanced whatever the rows hold; wasteful where rows
                                    are short): 0 = automatic (8 lanes, for every call that would
                                    otherwise get several lanes per target, i.e. small sets),
                                    -1 = never, else 4, 8, 16, 32 or 64                          */
    NBPP_OPT_P3M_ROWS_POOLED = 13, /* the row walk above with the pooled kernel's drain (survivors of
                                    a single-precision test pooled per wavefront, evaluated 64 at a
                                    time, sums gathered per target in LDS; not reproducible):
                                    0 = automatic (two lanes per target, for sets that fill the GPU
                                    with one lane per target and have begun to cluster), 1 = whenever
                                    NBPP_OPT_P3M_ROW_LANES asks for 2 to 16 lanes, -1 = never     */
    NBPP_OPT_P3M_POOL_WORKGROUP = 11 /* threads per workgroup of the pooled P3M kernel: 0 = 64, else
                                    64, 128 or 256.  Its wavefronts work independently, and alone
                                    on the GPU the kernel is 4-8 % faster with one wavefront per
                                    workgroup; a caller that runs other kernels beside it on a
                                    second stream should ask for 256, or single wavefront slots
                                    are refilled by this kernel before a larger workgroup of the
                                    other stream ever fits                                       */
};

/*
 * Lifecycle.  nbpp_init binds the calling process to HIP device `device` (-1 = current
 * device), creates the library's stream and events.  Idempotent; nbpp_shutdown frees
 * every device allocation the library made.
 * Replaces: the implicit device selection of the reference's `omp target` region
 * (src/particles.cpp:169).
 */
int nbpp_init(int device);
void nbpp_shutdown(void);
const char* nbpp_last_error(void);
int nbpp_set_option(int key, int64_t value);
int nbpp_get_stats(nbpp_stats* out);

/*
 * acc_{x,y,z}[i] += sum_{j != i} a_ij for i in [0, n).  HOST pointers, SoA fp64,
 * blocking, one caller thread.  This is the drop-in for
 *     ParticleSystem::compute_pp_forces(config)     src/particles.cpp:133
 * with x,y,z,m = pos_x/pos_y/pos_z/mass and ax,ay,az = acc_x/acc_y/acc_z
 * (src/particles.h:39-51).  The library owns all device memory.  n == 0 is a no-op.
 */
int nbpp_accumulate(const nbpp_params* p, size_t n,
                    const double* x, const double* y, const double* z, const double* m,
                    double* ax, double* ay, double* az);

/*
 * Optional, for callers of nbpp_accumulate that keep their arrays for the whole run (the
 * reference's std::vector members, src/particles.h:39-51): page-locked host memory for those
 * arrays (e.g. behind a std::vector allocator).  nbpp_accumulate recognises page-locked arrays by
 * itself (these, or anything from hipHostMalloc): positions and masses then travel by plain DMA,
 * and the device adds its sums to page-locked accumulators in place, instead of returning them
 * through a staging buffer and a host-side add.  Thin wrappers of hipHostMalloc / hipHostFree, so
 * that the caller does not need the HIP headers; nbpp_host_alloc returns NULL on failure
 * (nbpp_last_error() has the reason).  Replaces: nothing in the reference (its map(to/tofrom)
 * clauses, src/particles.cpp:169-173, leave the staging to the OpenMP runtime).
 */
void* nbpp_host_alloc(size_t bytes);
void nbpp_host_free(void* p);

/*
 * Device-resident form of the same operation (replaces the per-call map(to/tofrom)
 * re-mapping of src/particles.cpp:169-173).  x,y,z,m are DEVICE pointers to n_src
 * sources.  Targets are the sub-range [tgt_begin, tgt_begin + n_tgt) of the sources
 * (the i-partition of a multi-GPU run: each rank passes the all-gathered sources and
 * its own slice).  ax,ay,az are DEVICE pointers of length n_tgt, indexed by
 * i - tgt_begin, accumulated into.  Work is enqueued on `stream` (a hipStream_t; NULL is the
 * HIP null stream, as for every HIP call) and is ordered like any other work on that stream; the
 * host returns once the pair kernel is queued (the P3M path waits once in between, for the number
 * of cell-list entries), or when it has finished if NBPP_OPT_TIMING is on.
 * ax, ay, az may also be page-locked host memory that the device can reach (hipHostMalloc,
 * hipHostRegister): the library then keeps to the kernels that write their sums with plain
 * stores (the pair-once direct-summation kernel adds with fp64 atomics, which are only
 * guaranteed for ordinary device allocations, and is not chosen for such outputs).
 *
 * Concurrency.  The library's working memory (sorted copies, cell lists, queues, partial sums) is
 * one set per process.  Calls may come from several host threads (they are serialised on a lock
 * for the time it takes to queue them) and on several streams: a compute call whose predecessor
 * ran on a different stream first makes its stream wait for that predecessor's kernels
 * (hipStreamWaitEvent on an event the library records behind every compute call), so results are
 * correct on any mix of streams; the device work of two compute calls never overlaps.  Work of
 * OTHER kinds on other streams overlaps with a compute call as usual.  This holds for
 * nbpp_accumulate_device, nbpp_accumulate_pairs_device, nbpp_accumulate_device_noalloc and
 * nbpp_accumulate alike.  nbpp_get_stats describes the call that was queued last.
 */
int nbpp_accumulate_device(const nbpp_params* p, size_t n_src,
                           const double* x, const double* y, const double* z,
                           const double* m, size_t tgt_begin, size_t n_tgt,
                           double* ax, double* ay, double* az, void* stream);

/*
 * nbpp_accumulate_device over all targets (tgt_begin = 0, n_tgt = n) for callers that cannot
 * allocate device memory: code under HIP graph capture set-up, tight serving loops, harnesses that
 * forbid hipMalloc.  No counterpart in the reference.  The call makes no hipMalloc / hipFree and
 * calls no allocating library routine; the sorted copies, cell lists, queues and partial sums that
 * nbpp_accumulate_device keeps in buffers it allocates live here in one fixed array of
 * NBPP_NOALLOC_SCRATCH_BYTES inside the library's code object (every GPU the library runs on
 * holds that much from load time until the process ends).  Same kernels, same results, same
 * stream semantics and options as nbpp_accumulate_device.
 * Limits: n <= NBPP_NOALLOC_MAX_N, else NBPP_ERR_RANGE; parameters whose search grid does not
 * fit the array (a cutoff radius far below the particle spacing) also give NBPP_ERR_RANGE with
 * the sizes in the message, and nothing has been added to ax, ay, az then.
 * All calls on one GPU share the array and keep what the last call left in it; the library orders
 * them itself (see "Concurrency" above: a call on another stream waits for the one before it), so
 * two such calls never run at the same time.
 */
#define NBPP_NOALLOC_MAX_N 262144                      /* 64^3 particles */
#define NBPP_NOALLOC_SCRATCH_BYTES (48u * 1024u * 1024u)
int nbpp_accumulate_device_noalloc(const nbpp_params* p, size_t n,
                                   const double* x, const double* y, const double* z,
                                   const double* m, double* ax, double* ay, double* az,
                                   void* stream);

/*
 * One share of a direct-summation (use_pm == 0) evaluation, for multi-GPU runs that use every
 * pair once: the unordered pairs {i, j} of the n particles are divided into n_parts equal
 * shares; this call evaluates share `part` and adds BOTH a_i and a_j of each of its pairs into
 * ax, ay, az, which are DEVICE arrays of full length n in the caller's particle order.  The
 * sum of the arrays over all parts (e.g. an RCCL reduce-scatter) is the complete acceleration;
 * with n_parts == 1 the call is equivalent to nbpp_accumulate_device over all targets.
 * Uses fp64 atomics: not bitwise reproducible from run to run, and ax, ay, az must be ordinary
 * device allocations (hipMalloc and the like): host-mapped or managed memory is rejected with
 * NBPP_ERR_ARG, since hardware fp64 atomics are not guaranteed to land there.  No counterpart in
 * the reference, which evaluates every ordered pair (src/particles.cpp:270-365).
 */
int nbpp_accumulate_pairs_device(const nbpp_params* p, size_t n,
                                 const double* x, const double* y, const double* z,
                                 const double* m, int part, int n_parts,
                                 double* ax, double* ay, double* az, void* stream);

/*
 * Device-resident pieces of the caller contract around the hot path.  All pointers are
 * DEVICE pointers of length n; work is enqueued on `stream`.
 *
 * nbpp_kick_device      v += a*(dt/a^3) - v*(2 H dt)          src/integrator.cpp:168-181
 * nbpp_drift_device     x += v*dt; x -= L*floor(x/L)          src/integrator.cpp:183-198
 * nbpp_max_accel_sq_device
 *                       max(1e-9, max_i ax^2+ay^2+az^2) -> *out_host (blocking)
 *                                                              src/integrator.cpp:200-213
 * nbpp_add_device       a += b, per component                  src/particles.cpp:362-364
 *                       The final `acc[i] += sum` of compute_pp_forces for a caller that let the
 *                       short-range sums accumulate into arrays of their own (from zero) while
 *                       the mesh force was being computed on another stream: mesh + sum is the
 *                       same single rounding as the reference's mesh += sum.
 * nbpp_add_clear_device a += b, then b = 0: the same, leaving the sums' arrays ready for the
 *                       next evaluation without a separate pass over them.
 * nbpp_kick_drift_device       nbpp_kick_device(dt_kick) followed by nbpp_drift_device(dt_drift)
 * nbpp_add_clear_kick_device   nbpp_add_clear_device followed by nbpp_kick_device(dt)
 *                       in one pass over the particles each: the operations on every element and
 *                       their order are those of the separate calls, so are the results, bit for
 *                       bit (src/integrator.cpp:221-224 and :274 with the sum before it).
 * nbpp_add_clear_kick_max_device   nbpp_add_clear_kick_device followed by nbpp_max_accel_sq_device
 *                       on the new a (blocking, as that call is): the closing `acc += sum`, kick
 *                       and max |a|^2 of a cycle (src/particles.cpp:362-364, src/integrator.cpp:274,
 *                       :200-213) in one pass, the same value bit for bit.
 */
int nbpp_kick_device(size_t n, double* vx, double* vy, double* vz,
                     const double* ax, const double* ay, const double* az,
                     double dt, double scale_factor, double hubble, void* stream);
int nbpp_drift_device(size_t n, double* x, double* y, double* z,
                      const double* vx, const double* vy, const double* vz,
                      double dt, double domain_size, void* stream);
int nbpp_max_accel_sq_device(size_t n, const double* ax, const double* ay,
                             const double* az, double* out_host, void* stream);
int nbpp_add_device(size_t n, double* ax, double* ay, double* az,
                    const double* bx, const double* by, const double* bz, void* stream);
int nbpp_add_clear_device(size_t n, double* ax, double* ay, double* az,
                          double* bx, double* by, double* bz, void* stream);
int nbpp_kick_drift_device(size_t n, double* x, double* y, double* z,
                           double* vx, double* vy, double* vz,
                           const double* ax, const double* ay, const double* az,
                           double dt_kick, double dt_drift, double scale_factor, double hubble,
                           double domain_size, void* stream);
int nbpp_add_clear_kick_device(size_t n, double* ax, double* ay, double* az,
                               double* bx, double* by, double* bz,
                               double* vx, double* vy, double* vz,
                               double dt, double scale_factor, double hubble, void* stream);
int nbpp_add_clear_kick_max_device(size_t n, double* ax, double* ay, double* az,
                                   double* bx, double* by, double* bz,
                                   double* vx, double* vy, double* vz,
                                   double dt, double scale_factor, double hubble,
                                   double* out_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NBPP_H */
