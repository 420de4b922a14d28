// particles.hpp - host SoA dark-matter state with the reference's ParticleSystem interface for
// the particle-particle path (src/particles.h:26-72).  compute_pp_forces hands the arrays to
// libnbpp and libnbpm; there is no host implementation of the forces.
#pragma once
#include <cstddef>
#include <vector>

#include "config.hpp"

namespace nbh {

class ParticleSystem {
   public:
    std::size_t num_particles = 0;
    std::vector<double> pos_x, pos_y, pos_z;   // src/particles.h:39-41
    std::vector<double> vel_x, vel_y, vel_z;   // :43-45
    std::vector<double> acc_x, acc_y, acc_z;   // :47-49
    std::vector<double> mass;                  // :51
    double max_accel_sq = 0.0;                 // :53

    explicit ParticleSystem(const Config& config);

    // src/particles.cpp:32-45: appends one particle, acceleration zero.
    void add_particle(double px, double py, double pz, double vx, double vy, double vz, double m);

    // The reference builds its CIC density and its linked list here (src/particles.cpp:47-100).
    // libnbpp bins on the GPU inside compute_pp_forces and libnbpm assigns the density inside
    // compute_mesh_forces, so this only validates that positions are finite.
    void bin_and_assign_mass(const Config& config);

    // acc = mesh (PM) acceleration, on the GPU: density assignment, Poisson solve and
    // interpolate_cic_forces (src/particles.cpp:47-131, src/integrator.cpp:47-132) for dark
    // matter alone.  Throws std::runtime_error carrying nbpm_last_error() on failure.
    void compute_mesh_forces(const Config& config);

    // acc += PP / P3M short-range acceleration (src/particles.cpp:133-367), on the GPU.
    // Throws std::runtime_error carrying nbpp_last_error() on failure.
    void compute_pp_forces(const Config& config);

    // src/particles.cpp:386-391, as written there.
    double get_gravity_timestep(const Config& config) const;
};

}  // namespace nbh
