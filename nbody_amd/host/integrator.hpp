// integrator.hpp - the caller contract around compute_pp_forces, restating the dark-matter
// parts of the reference's KDK step (src/integrator.cpp:168-275): mesh force and/or
// particle-particle force, no gas.
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

// The force phase of a step (:236-269) for dark matter alone: the mesh force overwrites the
// accelerations if USE_PM, else they are zeroed (:242-251); the PP force is added if USE_PP
// (:254-258); then either max|a|^2 is reduced or, for STANDING_PARTICLES, the accelerations
// are erased (:261-269).
void compute_accelerations(ParticleSystem& dm, const Config& config);

// One kick-drift-kick step of the dark matter (:215-275 without the gas calls).
void KDK_step(ParticleSystem& dm, Cosmology& cosmo, double dt, const Config& config);

}  // namespace nbh
