/*
 * pp_oracle.h - CPU restatement of the reference's particle-particle gravity path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may load it,
 * and only as the checker or the timed CPU baseline.  libnbpp.so never links or
 * calls it.
 *
 * Parity pinning: oracle/_ref (built from the reference's own config.cpp and
 * math_utils.h) pins orc_config_derive and orc_periodic_displacement bit for bit.
 * The reference's particles.cpp cannot be built in this image (it needs
 * <Eigen/Dense>, which is not installed), so orc_pp_accumulate is pinned by the
 * known-answer vectors of the reference's own tests (tests/test_gravity.cpp:98-155,
 * tests/test_math.cpp:10-48) and by an independent long-double direct sum.
 */
#ifndef PP_ORACLE_H
#define PP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same layout as nbpp_params (include/nbpp.h); restated so that the oracle builds alone. */
typedef struct orc_params {
    double domain_size, G, soft_sq, pm_soft_sq;
    double r_switch, r_switch_sq, cutoff_sq, trans_width;
    double cell_size;
    int32_t mesh_size, use_pm;
} orc_params;

/* Inputs / outputs of Config::compute_derived_data (reference src/config.cpp:73-99). */
typedef struct orc_config_in {
    double domain_size;          /* DOMAIN_SIZE */
    int32_t mesh_size;           /* MESH_SIZE */
    int32_t n_per_side;          /* N_PER_SIDE */
    double omega_baryon, omega_m;
    double cutoff_radius_cells, cutoff_transition_width_factor;
    double dt_factor;
} orc_config_in;

typedef struct orc_config_out {
    double cell_size, cell_volume, omega_dm;
    double cutoff_radius, cutoff_radius_squared, cutoff_transition_width;
    double r_switch_start, r_switch_start_sq;
    int32_t num_dm_particles, pad;
    double G, dm_particle_mass, gas_total_mass, softening_squared, fixed_dt;
} orc_config_out;

enum {
    ORC_EXACT = 0,          /* every j != i exactly once, ascending j                        */
    ORC_REF_OFFLOAD = 1,    /* reference offload-branch traversal (particles.cpp:155-264):
                               linked list, -r..+r cell sweep with (c+d+N)%N aliasing, no cap */
    ORC_REF_CPU = 2,        /* reference CPU-branch traversal (particles.cpp:266-366): same
                               sweep, gather buffer silently truncated at 1500 neighbours    */
    ORC_EXACT_CELLS = 3     /* exact, but only the distinct cells within the search radius
                               are visited (use_pm only; fast at large n)                    */
};

double orc_periodic_displacement(double dx, double domain_size);
void orc_config_derive(const orc_config_in* in, orc_config_out* out);
void orc_params_from_config(const orc_config_in* in, int use_pm, orc_params* p);

/*
 * acc[i - tgt_begin] += sum over sources, for targets i in [tgt_begin, tgt_begin+n_tgt).
 * nthreads <= 0 uses every OpenMP thread.  Returns 0, or -1 on bad arguments.
 */
int orc_pp_accumulate(const orc_params* p, size_t n, const double* x, const double* y,
                      const double* z, const double* m, size_t tgt_begin, size_t n_tgt,
                      double* ax, double* ay, double* az, int traversal, int nthreads);

/* Same sum in long double with every j != i once: the independent cross-check. */
int orc_pp_accumulate_ld(const orc_params* p, size_t n, const double* x, const double* y,
                         const double* z, const double* m, size_t tgt_begin, size_t n_tgt,
                         double* ax, double* ay, double* az, int nthreads);

/* Ordered pairs (i in targets, j != i) with minimum-image r^2 <= cutoff_sq. */
uint64_t orc_count_within_cutoff(const orc_params* p, size_t n, const double* x,
                                 const double* y, const double* z, size_t tgt_begin,
                                 size_t n_tgt, int nthreads);

/* Caller-contract pieces (reference src/integrator.cpp:168-213, src/particles.cpp:386-391). */
void orc_kick(size_t n, double* vx, double* vy, double* vz, const double* ax,
              const double* ay, const double* az, double dt, double a, double H);
void orc_drift(size_t n, double* x, double* y, double* z, const double* vx,
               const double* vy, const double* vz, double dt, double domain_size);
double orc_max_accel_sq(size_t n, const double* ax, const double* ay, const double* az);
double orc_gravity_timestep(size_t n, double softening_squared, double max_accel_sq,
                            double gravity_dt_factor);
int orc_max_threads(void);

/* Pieces of the reference's Zel'dovich initial conditions (src/ics.cpp:36-165) that need C:
 * the libstdc++ white-noise stream and the sigma_8 normalisation loop.  The FFT part is in
 * oracle/zeldovich.py. */
void orc_libstdcxx_normal_field(uint32_t seed, size_t n, double* out);
double orc_bbks_pk(double k_h, double omega_m, double hubble, double n_s);
double orc_sigma8_amplitude(double sigma_8, double omega_m, double hubble, double n_s);
void orc_bbks_pk_array(size_t n, const double* k_h, double omega_m, double hubble, double n_s,
                       double* out);

#ifdef __cplusplus
}
#endif
#endif
