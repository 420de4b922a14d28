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

/*
 * Initial conditions: the Zel'dovich displacement field on the mesh and the particles displaced by
 * it, generated where the run will keep them.  Replaces compute_zeldovich_field after its white
 * noise and initialize_dm (src/ics.cpp:68-216); the white noise itself stays the caller's
 * (std::default_random_engine + std::normal_distribution on the host, src/ics.cpp:61-66: a stream
 * the C++ library defines, which a device generator could not reproduce).
 */
typedef struct nbpm_ic_params {
    int32_t mesh_size;      /* MESH_SIZE: cells per side of the displacement meshes          */
    int32_t n_per_side;     /* N_PER_SIDE: particles per side of the lattice                 */
    double domain_size;     /* DOMAIN_SIZE: the box in code units                            */
    double box_size_mpc;    /* BOX_SIZE_MPC: the same length in Mpc                          */
    double hubble_param;    /* h                                                             */
    double omega_m;         /* Omega_m (the spectrum's shape parameter is Omega_m h)         */
    double spectral_index;  /* n_s                                                           */
    double amplitude;       /* A of P(k) = A k^n T(k)^2: nbpm_sigma8_amplitude               */
    double scale_factor;    /* a at the start: the field is the displacement at that time    */
} nbpm_ic_params;

/*
 * *amplitude = sigma_8^2 / (variance of the unnormalised spectrum in spheres of 8 Mpc/h), with
 * the reference's own quadrature (src/ics.cpp:38-51: rectangle rule, k = 0.001 ... 100 h/Mpc in
 * steps of 0.001).  Host arithmetic; needs no GPU and no nbpm_init.
 */
int nbpm_sigma8_amplitude(double sigma_8, double omega_m, double hubble_param, double spectral_index,
                          double* amplitude);

/*
 * white_noise (mesh_size^3 unit Gaussians, DEVICE, preserved) -> dx, dy, dz (mesh_size^3 each,
 * DEVICE, overwritten): forward transform, one kernel over the mesh_size^2 (mesh_size/2 + 1) modes
 * (D_c = i c_c / c^2 sqrt(A P(k)) W, scaled to a discrete field and to the scale factor), three
 * inverse transforms.  Replaces src/ics.cpp:68-156.
 */
int nbpm_zeldovich_field_device(const nbpm_ic_params* p, const double* white_noise, double* dx, double* dy,
                                double* dz, void* stream);

/*
 * The n_per_side^3 particles: particle (a n + b) n + c starts at the centre of lattice cell
 * (a, b, c), is displaced by the field value of the mesh cell that centre lies in, folded into
 * [0, L), and gets the velocity velocity_factor * displacement (H f for Zel'dovich velocities,
 * 0 for particles at rest) and mass `mass`.  All arrays DEVICE; m may be NULL.
 * Replaces initialize_dm (src/ics.cpp:167-216).
 */
int nbpm_zeldovich_particles_device(const nbpm_ic_params* p, const double* dx, const double* dy, const double* dz,
                                    double velocity_factor, double mass, double* x, double* y, double* z, double* vx,
                                    double* vy, double* vz, double* m, void* stream);

#ifdef __cplusplus
}
#endif
#endif
