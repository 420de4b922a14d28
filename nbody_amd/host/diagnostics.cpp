#include "diagnostics.hpp"

#include <sys/resource.h>
#include <unistd.h>

#include <cmath>

namespace nbh {

Diagnostics compute_diagnostics(const Config& config, const ParticleSystem& dm,
                                const std::vector<double> (&gas)[9], const std::vector<double>& total_rho,
                                const std::vector<double>& phi, const Cosmology& cosmo, double dt_cfl,
                                double dt_gravity, double dt) {
    Diagnostics d;
    d.sim_time = cosmo.total_time;                                   // src/diagnostics.cpp:50-51
    d.scale_factor = cosmo.scale_factor;
    d.dt_cfl = dt_cfl;                                               // :54-56
    d.dt_gravity = dt_gravity;
    d.dt_final = dt;
    const bool hydro = config.USE_HYDRO && !gas[0].empty();
    const std::size_t cells = gas[0].size();
    if (hydro) {                                                     // :58-66
        for (std::size_t i = 0; i < cells; ++i) {
            if (i == 0 || gas[0][i] > d.max_gas_density) d.max_gas_density = gas[0][i];
            if (i == 0 || gas[5][i] > d.max_gas_pressure) d.max_gas_pressure = gas[5][i];
            const double v = std::sqrt(gas[6][i] * gas[6][i] + gas[7][i] * gas[7][i] + gas[8][i] * gas[8][i]);
            if (i == 0 || v > d.max_gas_velocity) d.max_gas_velocity = v;
        }
    }
    double sum_mass = 0.0, sum_p[3] = {0.0, 0.0, 0.0};                 // :69-79
    for (std::size_t i = 0; i < dm.num_particles; ++i) {
        sum_mass += dm.mass[i];
        sum_p[0] += dm.mass[i] * dm.vel_x[i];
        sum_p[1] += dm.mass[i] * dm.vel_y[i];
        sum_p[2] += dm.mass[i] * dm.vel_z[i];
    }
    d.total_mass_dm = sum_mass;
    for (int c = 0; c < 3; ++c) d.total_momentum_dm[c] = sum_p[c];
    if (config.ENABLE_ENERGY_DIAGNOSTICS) {                          // :87-100
        const double a = cosmo.scale_factor;                        // src/particles.cpp:369-384
        double ke = 0.0;
        for (std::size_t i = 0; i < dm.num_particles; ++i) {
            const double vx = dm.vel_x[i], vy = dm.vel_y[i], vz = dm.vel_z[i];
            const double proper_vel_sq = (a * vx) * (a * vx) + (a * vy) * (a * vy) + (a * vz) * (a * vz);
            ke += 0.5 * dm.mass[i] * proper_vel_sq;
        }
        d.ke_dm = ke;
        double pe = 0.0;
        if (total_rho.size() == phi.size())
            for (std::size_t i = 0; i < phi.size(); ++i) pe += 0.5 * total_rho[i] * phi[i] * config.CELL_VOLUME;
        d.pe_total = pe / a;
    }
    if (hydro) {                                                     // :103-128
        double mass = 0.0, mom[3] = {0.0, 0.0, 0.0}, ke = 0.0, ie = 0.0;
        for (std::size_t i = 0; i < cells; ++i) {
            mass += gas[0][i];
            mom[0] += gas[1][i]; mom[1] += gas[2][i]; mom[2] += gas[3][i];
            double k = 0.5 * (gas[1][i] * gas[1][i] + gas[2][i] * gas[2][i] + gas[3][i] * gas[3][i]);
            k = (gas[0][i] > 1e-12) ? k / gas[0][i] : 0.0;
            ke += k;
            ie += gas[5][i] / (config.GAMMA - 1.0);
        }
        d.total_mass_gas = mass * config.CELL_VOLUME;
        for (int c = 0; c < 3; ++c) d.total_momentum_gas[c] = mom[c];   // not scaled by the cell volume, as there
        d.ke_gas = ke * config.CELL_VOLUME;
        d.ie_gas = ie * config.CELL_VOLUME;
    }
    return d;
}

namespace {
double peak_memory_mb() {                                            // src/logger.cpp:13-20
    struct rusage usage;
    getrusage(RUSAGE_SELF, &usage);
    return static_cast<double>(usage.ru_maxrss) / 1024.0;
}
double current_memory_mb() {                                         // :23-41
    std::ifstream statm("/proc/self/statm");
    if (!statm.is_open()) return 0.0;
    long virtual_size = 0, rss_pages = 0;
    statm >> virtual_size >> rss_pages;
    return static_cast<double>(rss_pages * sysconf(_SC_PAGESIZE)) / (1024.0 * 1024.0);
}
}  // namespace

Logger::Logger(const std::string& run_dir) : start_(std::chrono::steady_clock::now()) {
    file_.open(run_dir + "/diagnostics.csv");                        // :54-55
}

void Logger::log(const Diagnostics& d) {
    if (!file_.is_open()) return;
    if (!header_written_) {                                          // :71-87
        file_ << "cycle,sim_time,scale_factor,"
              << "total_mass_gas,total_mass_dm,total_momentum_gas_x,total_momentum_gas_y,total_momentum_gas_z,"
                 "total_momentum_dm_x,total_momentum_dm_y,total_momentum_dm_z,"
              << "ke_gas,ke_dm,pe_dm,ie_gas,"
              << "dt_cfl,dt_gravity,dt_final,"
              << "max_gas_density,max_gas_pressure,max_gas_velocity,"
              << "wall_time_total,wall_time_pm,wall_time_pp,wall_time_hydro,"
                 "wall_time_io,cumulative_wall_time,memory_peak,memory_current\n";
        header_written_ = true;
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - start_).count();
    const int peak = static_cast<int>(peak_memory_mb()), current = static_cast<int>(current_memory_mb());
    file_ << d.cycle << "," << d.sim_time << "," << d.scale_factor << "," << d.total_mass_gas << ","   // :104-123
          << d.total_mass_dm << "," << d.total_momentum_gas[0] << "," << d.total_momentum_gas[1] << ","
          << d.total_momentum_gas[2] << "," << d.total_momentum_dm[0] << "," << d.total_momentum_dm[1] << ","
          << d.total_momentum_dm[2] << "," << d.ke_gas << "," << d.ke_dm << "," << d.pe_total << "," << d.ie_gas
          << "," << d.dt_cfl << "," << d.dt_gravity << "," << d.dt_final << "," << d.max_gas_density << ","
          << d.max_gas_pressure << "," << d.max_gas_velocity << "," << d.avg_step << "," << d.avg_pm << ","
          << d.avg_pp << "," << d.avg_hydro << "," << d.io_time << "," << wall << "," << peak << "," << current
          << "\n";
    file_.flush();
}

}  // namespace nbh
