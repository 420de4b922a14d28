/*
 * nbgas.h - C ABI of libnbgas.so: the gas side of the reference's step on one MI355X: the
 * first-order HLL finite-volume solver of src/gas.cpp (primitive variables, directional
 * splitting, floors, CFL time step), the gravity kick of the gas (src/integrator.cpp:134-166)
 * and its initial state (src/ics.cpp:241-288).  With libnbpm (mesh force) and libnbpp
 * (particle-particle force) it completes the reference's KDK step (src/integrator.cpp:215-275).
 *
 * All meshes are mesh_size^3 fp64 values in device memory, z fastest: value (i, j, k) sits at
 * (i * mesh_size + j) * mesh_size + k, the reference's Grid3D (src/types.h:17-19).  Entry
 * points take a hipStream_t (as void*), enqueue work and, unless stated, do not synchronise.
 * One caller thread at a time.
 */
#ifndef NBGAS_H
#define NBGAS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    NBGAS_OK = 0,
    NBGAS_ERR_ARG = -1,
    NBGAS_ERR_HIP = -2,
    NBGAS_ERR_NOINIT = -3,
    NBGAS_ERR_NOMEM = -5
};

typedef struct nbgas_params {
    double cell_size;         /* CELL_SIZE          (src/config.cpp:74) */
    double gamma;             /* GAMMA              (src/config.cpp:42) */
    double cfl_safety_factor; /* CFL_SAFETY_FACTOR  (src/config.cpp:55-56) */
    int32_t mesh_size;        /* MESH_SIZE */
    int32_t reserved;
} nbgas_params;

/* GasGrid's nine meshes (src/gas.h:46-48): conserved fields, then the primitive ones that
 * update_primitive_variables derives from them. */
typedef struct nbgas_state {
    double *density, *momentum_x, *momentum_y, *momentum_z, *energy;
    double *pressure, *velocity_x, *velocity_y, *velocity_z;
} nbgas_state;

int nbgas_init(int device);
void nbgas_shutdown(void);
const char* nbgas_last_error(void);

/* GasGrid::update_primitive_variables (src/gas.cpp:186-248): velocities and pressure from the
 * conserved fields, pressure floor 1e-12, energy re-synchronised with the floored pressure,
 * vacuum cells (density <= 1e-12). */
int nbgas_update_primitives_device(const nbgas_params* p, const nbgas_state* s, void* stream);

/* GasGrid::hydro_step (src/gas.cpp:250-299) with RiemannSolver::compute_fluxes / solve_hll
 * (:53-169): x, y and z sweeps of the first-order HLL scheme, density floor after every sweep,
 * primitive variables brought up to date at the end.  All nine meshes are updated in place. */
int nbgas_hydro_step_device(const nbgas_params* p, const nbgas_state* s, double dt, void* stream);

/* GasGrid::get_cfl_timestep (src/gas.cpp:301-323): cell_size / max(|v| + c_s) * safety factor,
 * +inf when the largest signal speed is below 1e-9.  Blocking: *dt_out is a host double. */
int nbgas_cfl_timestep_device(const nbgas_params* p, const nbgas_state* s, double* dt_out,
                              void* stream);
/* The same in two halves, for a caller that has other work to queue while the maximum is being
 * found: _enqueue_device queues the reduction and its read-back on `stream` and returns;
 * _finish waits for that read-back and gives the time step.  One enqueue in flight at a time. */
int nbgas_cfl_timestep_enqueue_device(const nbgas_params* p, const nbgas_state* s, void* stream);
int nbgas_cfl_timestep_finish(const nbgas_params* p, double* dt_out);

/* apply_gas_kick (src/integrator.cpp:134-166): primitives, then momentum += rho (g/a^3 - 2 H v) dt
 * and energy += v . (that source) dt.  gx, gy, gz: the mesh accelerations (libnbpm). */
int nbgas_kick_device(const nbgas_params* p, const nbgas_state* s, const double* gx,
                      const double* gy, const double* gz, double dt, double scale_factor,
                      double hubble, void* stream);
/* nbgas_kick_device followed by nbgas_cfl_timestep_enqueue_device in one pass over the meshes: the
 * reduction of get_cfl_timestep (src/gas.cpp:301-323) reads the pressure and velocities the kick
 * has just stored (src/integrator.cpp:139), so the kick hands them to it directly.  The time step
 * is collected with nbgas_cfl_timestep_finish, the same value as after the two separate calls. */
int nbgas_kick_cfl_enqueue_device(const nbgas_params* p, const nbgas_state* s, const double* gx,
                                  const double* gy, const double* gz, double dt,
                                  double scale_factor, double hubble, void* stream);

/* total = dm_density + gas density: the source of the mesh solve (src/integrator.cpp:57-59). */
int nbgas_total_density_device(const nbgas_params* p, const nbgas_state* s,
                               const double* dm_density, double* total, void* stream);

/* initialize_gas (src/ics.cpp:241-288): density = dm_density * mass_ratio floored at 1e-12,
 * velocity = hubble * displacement * growth_f (0 when `standing`, STANDING_PARTICLES),
 * energy = density * internal_energy + kinetic; then primitives.  disp_x/y/z: the Zel'dovich
 * displacement field on the mesh; internal_energy: get_internal_energy_from_temp_k (:218-239). */
int nbgas_initialize_device(const nbgas_params* p, const nbgas_state* s, const double* dm_density,
                            const double* disp_x, const double* disp_y, const double* disp_z,
                            double mass_ratio, double hubble, double growth_f, int standing,
                            double internal_energy, void* stream);

#ifdef __cplusplus
}
#endif
#endif
