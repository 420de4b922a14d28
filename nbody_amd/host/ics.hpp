// ics.hpp - Zel'dovich initial conditions for the dark matter (the reference's
// compute_zeldovich_field and initialize_dm, src/ics.cpp:36-216), generated on the GPU and left
// there.  Only two things happen on the host: the white noise, which must be the stream that
// std::default_random_engine + std::normal_distribution produce (src/ics.cpp:61-66), and the
// sigma_8 quadrature, a scalar.  The noise is uploaded once; libnbpm turns it into the three
// displacement meshes (one forward transform, one kernel over the modes, three inverse transforms)
// and places the particles (one kernel).  The engine takes both as device arrays.
#pragma once
#include <vector>

#include "config.hpp"
#include "device_array.hpp"
#include "integrator.hpp"
#include "particles.hpp"

namespace nbh {

struct ZeldovichField {              // src/ics.h, resident in HBM
    DeviceArray dx, dy, dz;          // MESH_SIZE^3 each, value (ix, iy, iz) at (ix M + iy) M + iz
    double f = 0.0;                  // growth rate Omega_m(a)^0.55 (src/ics.cpp:159-162)
};

struct DeviceParticles {             // what initialize_dm produces, resident in HBM
    DeviceArray x, y, z, vx, vy, vz, m;
    std::size_t count = 0;
};

ZeldovichField compute_zeldovich_field(double scale_factor, const Config& config);   // :36-165

// Lattice + displacement particles with Zel'dovich velocities (src/ics.cpp:167-216), on the device.
DeviceParticles initialize_dm_device(const Cosmology& cosmo, const Config& config, const ZeldovichField& zf);

// The same into a host ParticleSystem: the device arrays are copied back once and entered with
// add_particle, in the reference's particle order.
void initialize_dm(ParticleSystem& dm, const Cosmology& cosmo, const Config& config, const ZeldovichField& zf);

// The dark-matter part of initialize_state (src/ics.cpp:290-300): cosmology at t = 0, field,
// particles.  Accelerations are left at zero; the engine computes them.  `field_out`, if given,
// receives the displacement field, which the gas's initial state also needs (:302-303).
ParticleSystem initialize_particles(const Config& config, Cosmology& cosmo, ZeldovichField* field_out = nullptr);
DeviceParticles initialize_particles_device(const Config& config, Cosmology& cosmo, ZeldovichField* field_out = nullptr);

// get_internal_energy_from_temp_k (src/ics.cpp:218-239): specific internal energy in code units.
double internal_energy_from_temperature(double T_kelvin, double hubble_param, double box_size_mpc,
                                        double domain_size, double gamma);

}  // namespace nbh
