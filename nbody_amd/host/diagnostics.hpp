// diagnostics.hpp - the conservation / stability numbers the reference logs every
// DEBUG_INFO_EVERY_CYCLES cycles (Diagnostics::update_physics, src/diagnostics.cpp:47-129) and
// its CSV log (Logger, src/logger.cpp:52-123: run_dir/diagnostics.csv, same columns in the same
// order).  Computed on the host from a downloaded copy of the engine's state: logging is rare.
#pragma once
#include <chrono>
#include <fstream>
#include <string>
#include <vector>

#include "config.hpp"
#include "integrator.hpp"
#include "particles.hpp"

namespace nbh {

struct Diagnostics {                       // src/diagnostics.h:11-33
    long cycle = 0;                        // the reference never updates its counter and logs 0;
                                           // the cycle number is written here
    double sim_time = 0.0, scale_factor = 1.0;
    double total_mass_gas = 0.0, total_mass_dm = 0.0;
    double total_momentum_gas[3] = {0.0, 0.0, 0.0}, total_momentum_dm[3] = {0.0, 0.0, 0.0};
    double ke_gas = 0.0, ke_dm = 0.0, pe_total = 0.0, ie_gas = 0.0;
    double dt_cfl = 0.0, dt_gravity = 0.0, dt_final = 0.0;
    double max_gas_density = 0.0, max_gas_pressure = 0.0, max_gas_velocity = 0.0;
    // seconds per cycle since the previous log line: whole step, mesh force, PP force, gas
    double avg_step = 0.0, avg_pm = 0.0, avg_pp = 0.0, avg_hydro = 0.0, io_time = 0.0;

    double total_mass() const { return total_mass_dm + total_mass_gas; }               // :39-41
    double total_energy() const { return ke_dm + ke_gas + ie_gas + pe_total; }         // :42-44
};

// update_physics for a state snapshot.  gas: GasGrid's nine meshes (empty without USE_HYDRO);
// total_rho, phi: source and (unnormalised) potential of the last mesh solve (may be empty:
// then the potential energy is 0).
Diagnostics compute_diagnostics(const Config& config, const ParticleSystem& dm,
                                const std::vector<double> (&gas)[9], const std::vector<double>& total_rho,
                                const std::vector<double>& phi, const Cosmology& cosmo, double dt_cfl,
                                double dt_gravity, double dt);

class Logger {
   public:
    explicit Logger(const std::string& run_dir);
    void log(const Diagnostics& d);      // one CSV row (header first)
    bool is_open() const { return file_.is_open(); }

   private:
    std::ofstream file_;
    bool header_written_ = false;
    std::chrono::steady_clock::time_point start_;
};

}  // namespace nbh
