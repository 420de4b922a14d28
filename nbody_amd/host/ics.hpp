// ics.hpp - Zel'dovich initial conditions for the dark matter, restating the reference's
// compute_zeldovich_field and initialize_dm (src/ics.cpp:10-216): BBKS spectrum normalised to
// sigma_8, Gaussian white noise from std::default_random_engine + std::normal_distribution
// (the same standard-library stream the reference draws), displacement field by FFT, particles on
// a lattice displaced by the field.  The three-dimensional transforms run on the GPU (hipFFT).
#pragma once
#include <vector>

#include "config.hpp"
#include "integrator.hpp"
#include "particles.hpp"

namespace nbh {

struct ZeldovichField {          // src/ics.h
    std::vector<double> dx, dy, dz;   // MESH_SIZE^3 each, index (ix*M + iy)*M + iz
    double f = 0.0;                   // growth rate Omega_m(a)^0.55
};

ZeldovichField compute_zeldovich_field(double scale_factor, const Config& config);   // :36-165

// Lattice + displacement particles with Zel'dovich velocities (src/ics.cpp:167-216).
void initialize_dm(ParticleSystem& dm, const Cosmology& cosmo, const Config& config,
                   const ZeldovichField& zf);

// The dark-matter part of initialize_state (src/ics.cpp:290-300): cosmology at t = 0, field,
// particles.  Accelerations are left at zero; the engine computes them.  `field_out`, if given,
// receives the displacement field, which the gas's initial state also needs (:302-303).
ParticleSystem initialize_particles(const Config& config, Cosmology& cosmo, ZeldovichField* field_out = nullptr);

// get_internal_energy_from_temp_k (src/ics.cpp:218-239): specific internal energy in code units.
double internal_energy_from_temperature(double T_kelvin, double hubble_param, double box_size_mpc,
                                        double domain_size, double gamma);

}  // namespace nbh
