/*
 * nbpm.h - C ABI of libnbpm.so: the mesh (particle-mesh, long-range) side of the reference's
 * gravity step on one MI355X: cloud-in-cell mass assignment, FFT Poisson solve, central
 * differences, cloud-in-cell interpolation.  It is the block the reference's KDK step runs
 * right before compute_pp_forces (src/integrator.cpp:236-252), whose result the
 * particle-particle force of libnbpp (include/nbpp.h) is then accumulated onto.
 *
 * All arrays are fp64.  Meshes are mesh_size^3 values, z fastest: value (i, j, k) sits at
 * (i * mesh_size + j) * mesh_size + k, the reference's Grid3D (src/types.h:17-19).
 * Device entry points take device pointers and a hipStream_t (as void*); they enqueue work and
 * return without synchronising.  One caller thread at a time.
 */
#ifndef NBPM_H
#define NBPM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    NBPM_OK = 0,
    NBPM_ERR_ARG = -1,    /* bad argument */
    NBPM_ERR_HIP = -2,    /* HIP or hipFFT failure */
    NBPM_ERR_NOINIT = -3, /* nbpm_init not called */
    NBPM_ERR_NOMEM = -5   /* device allocation failed */
};

typedef struct nbpm_params {
    double domain_size; /* DOMAIN_SIZE  (src/config.h) */
    double G;           /* G            (src/config.cpp:86) */
    int32_t mesh_size;  /* MESH_SIZE: cells per side */
    int32_t reserved;
} nbpm_params;

typedef struct nbpm_stats {
    double deposit_ms, solve_ms, interpolate_ms, total_ms; /* device time of the last call with
                                                              timing on (nbpm_set_timing) */
} nbpm_stats;

int nbpm_init(int device);
void nbpm_shutdown(void);
const char* nbpm_last_error(void);
int nbpm_set_timing(int on);
int nbpm_get_stats(nbpm_stats* out);

/*
 * rho(cell) = sum of m * w over the particles whose cloud touches the cell, / cell volume.
 * Replaces the density part of ParticleSystem::bin_and_assign_mass
 * (src/particles.cpp:47-93, :99).  `rho` (mesh_size^3) is overwritten.  Sums are formed with
 * fp64 atomics: the order of additions, and so the last bits, vary from run to run.
 */
int nbpm_deposit_device(const nbpm_params* p, size_t n, const double* x, const double* y,
                        const double* z, const double* m, double* rho, void* stream);

/*
 * Mesh accelerations from the total density: rho -> r2c FFT -> phi_k = -4 pi G rho_k / k^2
 * (phi_0 = 0) -> c2r FFT -> central differences with factor -1 / (N^3 2 cell).
 * Replaces compute_gravitational_acceleration after the density sum
 * (src/integrator.cpp:61-131).  gx, gy, gz (mesh_size^3 each) are overwritten; `rho` is
 * preserved.
 */
int nbpm_solve_device(const nbpm_params* p, const double* rho, double* gx, double* gy,
                      double* gz, void* stream);

/*
 * Copies into `phi` (mesh_size^3, device) the potential mesh the last nbpm_solve_device (or
 * nbpm_accelerations*) of that mesh size computed: the unnormalised inverse transform, i.e. the
 * reference's state.phi (src/integrator.cpp:105-106), which its energy diagnostics read
 * (src/diagnostics.cpp:93-97).
 */
int nbpm_potential_device(const nbpm_params* p, double* phi, void* stream);

/*
 * acc[i] = cloud-in-cell interpolation of the three meshes at particle i (overwrite).
 * Replaces ParticleSystem::interpolate_cic_forces (src/particles.cpp:102-131); the weights
 * the reference caches in cic_data at assignment time are recomputed from the positions,
 * which do not change in between (src/integrator.cpp:239-245).
 */
int nbpm_interpolate_device(const nbpm_params* p, size_t n, const double* x, const double* y,
                            const double* z, const double* gx, const double* gy,
                            const double* gz, double* ax, double* ay, double* az,
                            void* stream);

/*
 * The three calls above on library-owned meshes: what the reference's step does at
 * src/integrator.cpp:239-245 for dark matter alone (USE_HYDRO off).  acc is overwritten.
 */
int nbpm_accelerations_device(const nbpm_params* p, size_t n, const double* x, const double* y,
                              const double* z, const double* m, double* ax, double* ay,
                              double* az, void* stream);

/* Same with host pointers; blocking. */
int nbpm_accelerations(const nbpm_params* p, size_t n, const double* x, const double* y,
                       const double* z, const double* m, double* ax, double* ay, double* az);

#ifdef __cplusplus
}
#endif
#endif
