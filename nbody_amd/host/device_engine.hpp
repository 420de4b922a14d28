// device_engine.hpp - the simulation loop with all state resident in HBM: the reference's
// SimulationEngine (src/engine.cpp:10-91) and KDK_step (src/integrator.cpp:215-275), driving the
// device entry points of libnbpp (particle-particle force, kick, drift), libnbpm (mesh force)
// and libnbgas (gas).  Positions, velocities, accelerations and the gas meshes cross PCIe once in
// each direction (upload at construction, download on request) instead of twice per step.
// No snapshots, no logging.
//
// Several GPUs: when libnbmg has been initialised before the engine is built (nbmg_init /
// nbmg_init_from_env, one process per GPU), every process constructs the engine from the SAME
// initial state and keeps only its own block of the particles (nbmg_partition) in HBM; each force
// phase all-gathers the positions (RCCL) inside nbmg_accumulate_device, the dark-matter density is
// deposited per block and summed over the ranks, and the mesh solve and the gas, which are small
// next to the pair force, are computed redundantly on every rank from identical inputs.  step(),
// download() and the time-step functions are then collective: every rank must call them in the
// same order.
#pragma once
#include <cstddef>
#include <vector>

#include "config.hpp"
#include "ics.hpp"
#include "integrator.hpp"
#include "particles.hpp"

namespace nbh {

enum class ExitStatus { ReachedMaxScaleFactor, ReachedMaxCycles, UserAborted };  // src/engine.h

class DeviceEngine {
   public:
    // Uploads the host state and computes the initial accelerations and time step, as the
    // reference's constructor does through initialize_state (src/engine.cpp:16-22).  With
    // USE_HYDRO the gas starts from the displacement field `field` (src/ics.cpp:241-288), which
    // must then be given.
    DeviceEngine(const Config& config, const ParticleSystem& initial, const ZeldovichField* field = nullptr);
    // The same from initial conditions that were generated on the device (initialize_particles_device):
    // nothing passes through host memory.
    DeviceEngine(const Config& config, const DeviceParticles& initial, const ZeldovichField* field = nullptr);
    ~DeviceEngine();
    DeviceEngine(const DeviceEngine&) = delete;
    DeviceEngine& operator=(const DeviceEngine&) = delete;

    void step();                         // src/engine.cpp:36-54 (no I/O, no logging)
    ExitStatus run();                    // src/engine.cpp:78-91
    void request_stop() { stop_requested_ = true; }
    void download(ParticleSystem& out);  // blocking copy of the particle state back to the host
    // blocking copy of the nine gas meshes (density, momentum x/y/z, energy, pressure,
    // velocity x/y/z; MESH_SIZE^3 each); empty without USE_HYDRO
    void download_gas(std::vector<double> (&out)[9]);
    // blocking copy of the source and the (unnormalised) potential of the mesh solve, the
    // reference's state.total_rho and state.phi.  When the step itself uses no mesh (pair force
    // only, no gas) they are computed here, from the current positions, as the reference computes
    // them on every cycle whatever the switches (src/integrator.cpp:239-240).  Collective with
    // several ranks.
    void download_mesh_solve(std::vector<double>& total_rho, std::vector<double>& phi);
    double cfl_timestep();               // src/gas.cpp:301-323; +inf without USE_HYDRO
    double gravity_timestep() const;     // src/particles.cpp:386-391

    double current_dt = 0.0;
    long cycle_count = 0;
    bool needs_more_cycles = true;
    Cosmology cosmo;
    double max_accel_sq = 1e-9;
    // Device times are measured with events, and reading an event means waiting for the GPU: on
    // small problems that wait costs more than the kernels it times.  Cycles whose number is a
    // multiple of timing_every are timed (1: every cycle, 0: none); the sums below cover those only.
    int timing_every = 1;
    long timed_force_calls = 0;          // force phases (the initial one included) the sums cover
    long timed_steps = 0;                // cycles gas_ms covers
    double pp_ms = 0.0;                  // device time of the timed PP force calls
    double pm_ms = 0.0;                  // device time of the timed mesh force calls
    double gas_ms = 0.0;                 // device time of the timed gas sweeps
    double comm_ms = 0.0;                // part of pp_ms spent in the position all-gathers and the
                                         // reduce-scatters (several GPUs)
    // On cycles that are not timed, the short-range force runs on a stream of its own (into
    // arrays of its own, from zero) while the gas sweeps and the mesh force run on the main one;
    // the two are added at the end, which rounds exactly as accumulating onto the mesh force does
    // (src/particles.cpp:362-364).  The pair kernel is bound by instruction issue and ends in a
    // tail of few long wavefronts; the mesh and gas kernels are bound by memory and fill in.
    bool overlap_forces = true;

   private:
    void construct(std::size_t n_total, const double* const (&state)[10], const ZeldovichField* field);
    void compute_accelerations();        // src/integrator.cpp:236-269
    void deposit_density();
    bool launch_short_range();           // the overlapped short-range force, on its own stream
    double next_timestep();              // src/engine.cpp:16-22, :47-51
    nbgas_state gas_state() const;
    Config config_;
    std::size_t n_ = 0, cells_ = 0;      // particles held by this process, mesh cells
    std::size_t n_total_ = 0, begin_ = 0;   // all particles; index of this process's first one
    int ranks_ = 0;                      // 0: libnbmg not in use; else the number of processes
    double* dev_[10] = {};               // x y z vx vy vz ax ay az (own block) and m (ALL particles)
    double* gather_ = nullptr;           // several ranks: one padded all-particle array for download()
    double* phi_stage_ = nullptr;        // download_mesh_solve: the potential on its way to the host
    bool meshes_on_demand_ = false;      // no mesh in the step (pair force only, no gas): density and
                                         // potential are computed when download_mesh_solve asks
    double* gas_[9] = {};                // GasGrid's meshes, in nbgas_state order
    double* mesh_[5] = {};               // dm density, total density, gravity x y z
    bool stop_requested_ = false;
    bool timed_ = true;                  // the force phase / cycle in flight is being timed
    void* ev_[4] = {};                   // hipEvent_t: mesh begin/end, gas begin/end
    void* main_ = nullptr;               // hipStream_t (non-blocking) every call of the step is queued on
    void* pp_stream_ = nullptr;          // hipStream_t (non-blocking) of the overlapped short-range force
    void* ev_pos_ = nullptr;             // positions of this cycle are final (recorded after the drift)
    void* ev_pp_ = nullptr;              // short-range sums complete
    double gas_kick_after_mesh_ = 0.0;   // > 0: compute_accelerations() queues the gas's half kick of this length
    double kick_after_forces_ = 0.0;     // > 0: compute_accelerations() may kick the particles by this much with its sum
    bool kick_fused_ = false;            // ... and has done so
    bool max_fused_ = false;             // ... together with max |a|^2 of the sum
    bool cfl_in_flight_ = false;         // the gas's CFL limit has been queued (nbgas_cfl_timestep_enqueue_device)
    bool pp_acc_zeroed_ = false;         // pp_acc_ holds zeros for the next overlapped evaluation
    bool pos_event_recorded_ = false;
    double last_occupancy_ = 0.0;        // clustering measure of the last short-range call (nbpp_stats)
    bool pp_launched_ = false;           // step() has started this cycle's short-range force already
    double* pp_acc_[3] = {};             // short-range sums of the overlapped path
};

}  // namespace nbh
