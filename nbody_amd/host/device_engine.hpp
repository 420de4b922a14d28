// device_engine.hpp - the simulation loop for dark matter with all particle state resident
// in HBM: the reference's SimulationEngine (src/engine.cpp:10-91) and KDK_step
// (src/integrator.cpp:215-275) reduced to the dark-matter parts (mesh force through libnbpm,
// particle-particle force through libnbpp; no gas), driving the libraries' device entry points.  Positions, velocities and accelerations cross PCIe once in
// each direction (upload at construction, download on request) instead of twice per step.
#pragma once
#include <cstddef>

#include "config.hpp"
#include "integrator.hpp"
#include "particles.hpp"

namespace nbh {

enum class ExitStatus { ReachedMaxScaleFactor, ReachedMaxCycles, UserAborted };  // src/engine.h

class DeviceEngine {
   public:
    // Uploads the host state and computes the initial accelerations and time step, as the
    // reference's constructor does through initialize_state (src/engine.cpp:16-22).
    DeviceEngine(const Config& config, const ParticleSystem& initial);
    ~DeviceEngine();
    DeviceEngine(const DeviceEngine&) = delete;
    DeviceEngine& operator=(const DeviceEngine&) = delete;

    void step();                         // src/engine.cpp:36-54 (no I/O, no logging)
    ExitStatus run();                    // src/engine.cpp:78-91
    void request_stop() { stop_requested_ = true; }
    void download(ParticleSystem& out);  // blocking copy of the full state back to the host

    double current_dt = 0.0;
    long cycle_count = 0;
    bool needs_more_cycles = true;
    Cosmology cosmo;
    double max_accel_sq = 1e-9;
    double pp_ms = 0.0;                  // accumulated device time of the PP force calls
    double pm_ms = 0.0;                  // accumulated device time of the mesh force calls

   private:
    void compute_accelerations();        // src/integrator.cpp:236-269 for dark matter alone
    double gravity_timestep() const;     // src/particles.cpp:386-391
    Config config_;
    std::size_t n_ = 0;
    double* dev_[10] = {};               // x y z vx vy vz ax ay az m
    bool stop_requested_ = false;
};

}  // namespace nbh
