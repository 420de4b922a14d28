#include "integrator.hpp"

#include <algorithm>
#include <cmath>

namespace nbh {

void update_cosmology(Cosmology& c, const Config& config) {
    if (!config.EXPANDING_UNIVERSE) {                       // src/integrator.cpp:10-14
        c.scale_factor = 1.0;
        c.hubble_param = 0.0;
        return;
    }
    if (config.OMEGA_LAMBDA == 0.0) {                       // Einstein-de Sitter, :16-21
        const double t = std::pow(config.START_A, 1.5) + c.total_time;
        c.scale_factor = std::pow(t, 2.0 / 3.0);
        c.hubble_param = (2.0 / 3.0) / t;
        return;
    }
    const double Om = config.OMEGA_M, Ol = config.OMEGA_LAMBDA;   // Lambda-CDM, :23-44
    const double H0 = 2.0 / (3.0 * std::sqrt(Om));
    const double factor = std::sqrt(Ol / Om);
    const double t_start =
        (2.0 / (3.0 * H0 * std::sqrt(Ol))) * std::asinh(factor * std::pow(config.START_A, 1.5));
    const double t = t_start + c.total_time;
    const double sinh_term = std::sinh(1.5 * H0 * std::sqrt(Ol) * t);
    c.scale_factor = std::pow(Om / Ol, 1.0 / 3.0) * std::pow(sinh_term, 2.0 / 3.0);
    const double a3 = std::pow(c.scale_factor, 3.0);
    c.hubble_param = H0 * std::sqrt(Om / a3 + Ol);
}

void apply_dm_kick(ParticleSystem& dm, double dt, double a, double H) {
    const double a3 = a * a * a;
    const double dt_over_a3 = dt / a3;
    const double dt_2H = dt * 2.0 * H;
    for (std::size_t i = 0; i < dm.num_particles; ++i) {
        dm.vel_x[i] += dm.acc_x[i] * dt_over_a3 - dm.vel_x[i] * dt_2H;
        dm.vel_y[i] += dm.acc_y[i] * dt_over_a3 - dm.vel_y[i] * dt_2H;
        dm.vel_z[i] += dm.acc_z[i] * dt_over_a3 - dm.vel_z[i] * dt_2H;
    }
}

void apply_dm_drift(ParticleSystem& dm, double dt, double domain_size) {
    const double inv_domain = 1.0 / domain_size;
    for (std::size_t i = 0; i < dm.num_particles; ++i) {
        const double nx = dm.pos_x[i] + dm.vel_x[i] * dt;
        const double ny = dm.pos_y[i] + dm.vel_y[i] * dt;
        const double nz = dm.pos_z[i] + dm.vel_z[i] * dt;
        dm.pos_x[i] = nx - domain_size * std::floor(nx * inv_domain);
        dm.pos_y[i] = ny - domain_size * std::floor(ny * inv_domain);
        dm.pos_z[i] = nz - domain_size * std::floor(nz * inv_domain);
    }
}

void calculate_max_acceleration(ParticleSystem& dm) {
    double local_max = 1e-9;
    for (std::size_t i = 0; i < dm.num_particles; ++i) {
        const double s = dm.acc_x[i] * dm.acc_x[i] + dm.acc_y[i] * dm.acc_y[i] + dm.acc_z[i] * dm.acc_z[i];
        local_max = (s > local_max) ? s : local_max;
    }
    dm.max_accel_sq = local_max;
}

void compute_accelerations(ParticleSystem& dm, const Config& config) {
    dm.bin_and_assign_mass(config);
    if (config.USE_PM) {
        dm.compute_mesh_forces(config);   // overwrites acc, as interpolate_cic_forces does
    } else {
        std::fill(dm.acc_x.begin(), dm.acc_x.end(), 0.0);
        std::fill(dm.acc_y.begin(), dm.acc_y.end(), 0.0);
        std::fill(dm.acc_z.begin(), dm.acc_z.end(), 0.0);
    }
    if (config.USE_PP) dm.compute_pp_forces(config);
    if (!config.STANDING_PARTICLES) {
        calculate_max_acceleration(dm);
    } else {
        std::fill(dm.acc_x.begin(), dm.acc_x.end(), 0.0);
        std::fill(dm.acc_y.begin(), dm.acc_y.end(), 0.0);
        std::fill(dm.acc_z.begin(), dm.acc_z.end(), 0.0);
        dm.max_accel_sq = 1e-9;
    }
}

void KDK_step(ParticleSystem& dm, Cosmology& cosmo, double dt, const Config& config) {
    update_cosmology(cosmo, config);
    apply_dm_kick(dm, dt / 2.0, cosmo.scale_factor, cosmo.hubble_param);
    apply_dm_drift(dm, dt, config.DOMAIN_SIZE);
    cosmo.total_time += dt;
    update_cosmology(cosmo, config);
    compute_accelerations(dm, config);
    apply_dm_kick(dm, dt / 2.0, cosmo.scale_factor, cosmo.hubble_param);
}

}  // namespace nbh
