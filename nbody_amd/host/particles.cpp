#include "particles.hpp"

#include <cmath>
#include <limits>
#include <stdexcept>
#include <string>

namespace nbh {

ParticleSystem::ParticleSystem(const Config& config) {
    const std::size_t cap = config.NUM_DM_PARTICLES > 0 ? config.NUM_DM_PARTICLES : 0;
    for (auto* v : {&pos_x, &pos_y, &pos_z, &vel_x, &vel_y, &vel_z, &acc_x, &acc_y, &acc_z, &mass})
        v->reserve(cap);
}

void ParticleSystem::add_particle(double px, double py, double pz, double vx, double vy, double vz,
                                  double m) {
    pos_x.push_back(px); pos_y.push_back(py); pos_z.push_back(pz);
    vel_x.push_back(vx); vel_y.push_back(vy); vel_z.push_back(vz);
    acc_x.push_back(0.0); acc_y.push_back(0.0); acc_z.push_back(0.0);
    mass.push_back(m);
    ++num_particles;
}

void ParticleSystem::bin_and_assign_mass(const Config&) {
    for (std::size_t i = 0; i < num_particles; ++i)
        if (!std::isfinite(pos_x[i]) || !std::isfinite(pos_y[i]) || !std::isfinite(pos_z[i]))
            throw std::runtime_error("non-finite particle position at index " + std::to_string(i));
}

void ParticleSystem::compute_pp_forces(const Config& config) {
    const nbpp_params p = config.pp_params();
    const int rc = nbpp_accumulate(&p, num_particles, pos_x.data(), pos_y.data(), pos_z.data(),
                                   mass.data(), acc_x.data(), acc_y.data(), acc_z.data());
    if (rc != NBPP_OK) throw std::runtime_error(std::string("libnbpp: ") + nbpp_last_error());
}

void ParticleSystem::compute_mesh_forces(const Config& config) {
    const nbpm_params p = config.mesh_params();
    if (nbpm_init(0) != NBPM_OK) throw std::runtime_error(std::string("libnbpm: ") + nbpm_last_error());
    const int rc = nbpm_accelerations(&p, num_particles, pos_x.data(), pos_y.data(), pos_z.data(),
                                      mass.data(), acc_x.data(), acc_y.data(), acc_z.data());
    if (rc != NBPM_OK) throw std::runtime_error(std::string("libnbpm: ") + nbpm_last_error());
}

double ParticleSystem::get_gravity_timestep(const Config& config) const {
    if (num_particles == 0) return std::numeric_limits<double>::infinity();
    const double dt_grav = std::sqrt(config.SOFTENING_SQUARED) / std::sqrt(max_accel_sq);
    return dt_grav * config.GRAVITY_DT_FACTOR;
}

}  // namespace nbh
