// integrator.hpp - the caller contract around compute_pp_forces, restating the dark-matter
// parts of the reference's KDK step (src/integrator.cpp:168-275) for PP-only gravity.
#pragma once
#include "config.hpp"
#include "particles.hpp"

namespace nbh {

// Scale factor and Hubble rate in code units (src/integrator.cpp:9-45).
struct Cosmology {
    double total_time = 0.0;
    double scale_factor = 1.0;
    double hubble_param = 0.0;
};
void update_cosmology(Cosmology& c, const Config& config);

void apply_dm_kick(ParticleSystem& dm, double dt, double a, double H);     // :168-181
void apply_dm_drift(ParticleSystem& dm, double dt, double domain_size);    // :183-198
void calculate_max_acceleration(ParticleSystem& dm);                       // :200-213

// The force phase of a step with the mesh (PM) side absent: zero the accelerations
// (:246-251), add the PP force if USE_PP (:254-258), then either reduce max|a|^2 or, for
// STANDING_PARTICLES, erase the accelerations (:261-269).
void compute_pp_accelerations(ParticleSystem& dm, const Config& config);

// One kick-drift-kick step of the dark matter with PP-only gravity (:215-275 without the gas
// and mesh calls).
void KDK_step(ParticleSystem& dm, Cosmology& cosmo, double dt, const Config& config);

}  // namespace nbh
