// simulation.hpp - one whole run: initial conditions, the device-resident engine, snapshots and
// diagnostics.csv with the reference's cadence.  The counterpart of the reference's
// SimulationEngine constructor + run() (src/engine.cpp:10-34, :36-91) around DeviceEngine, which
// is the KDK step alone.
#pragma once
#include <chrono>
#include <csignal>
#include <memory>
#include <string>
#include <vector>

#include "config.hpp"
#include "device_engine.hpp"
#include "diagnostics.hpp"
#include "ics.hpp"
#include "snapshot.hpp"

namespace nbh {

struct RunOptions {
    // Where snapshots go (cadence: one at the start, one whenever the scale factor passes the next
    // multiple of SAVE_HDF5_EVERY_DELTA_A, one at the end; src/engine.cpp:28-33, :56-67).
    // A directory: snapshot_%04d.hdf5 files in the reference's layout (snapshot.hpp).  Anything
    // else is a file prefix for raw little-endian doubles, <prefix>_%04d.bin:
    // {n, mesh, a, t, L, use_hydro}, x, y, z, mass[n], then gas density and energy [mesh^3] with
    // hydro.  Empty: no snapshots.
    std::string snapshot_target;
    // Directory of diagnostics.csv (the reference's columns and cadence, src/engine.cpp:24-26,
    // :69-75); empty: none.
    std::string log_dir;
    // Cycles between device-time samples (see DeviceEngine::timing_every); negative: on the cycles
    // that report, and on the last one of a run bounded by its cycle count.
    int timing_every = -1;
    // "cycle N: ..." progress lines every DEBUG_INFO_EVERY_CYCLES cycles
    bool progress = true;
    // DeviceEngine::overlap_forces
    bool overlap_forces = true;
};

class Simulation {
   public:
    // Builds the Zel'dovich initial conditions of `config`, uploads them and computes the
    // initial forces.  nbpp_init / nbpm_init / nbgas_init must have been called (DeviceEngine does).
    Simulation(const Config& config, const RunOptions& options);

    // Steps until MAX_SCALE_FACTOR or MAX_CYCLES is reached or *stop becomes non-zero (checked
    // once per cycle; may be null).  Ctrl+C handlers set it: the cycle in flight is finished and
    // the last snapshot and diagnostics line are written (src/main.cpp:13-22, src/engine.cpp:79).
    ExitStatus run(const volatile std::sig_atomic_t* stop);

    // "<when>: cycle ... / p0 acc ... / gas mass ..." on stdout, full precision
    void report(const char* when);
    void print_timing(ExitStatus status) const;

    DeviceEngine& engine() { return *engine_; }
    int snapshots_written() const { return snapshot_count_; }

   private:
    void save_snapshot();
    void log_diagnostics();

    Config config_;
    RunOptions options_;
    std::unique_ptr<DeviceEngine> engine_;
    std::unique_ptr<SnapshotWriter> hdf5_;
    std::unique_ptr<Logger> logger_;
    int snapshot_count_ = 0;
    double next_output_a_ = 0.0;
    long logged_cycle_ = 0, logged_calls_ = 0, logged_steps_ = 0;
    double logged_ms_[3] = {0.0, 0.0, 0.0}, last_avg_[3] = {0.0, 0.0, 0.0};
    std::chrono::steady_clock::time_point logged_wall_;
    double run_wall_ms_ = 0.0;
};

}  // namespace nbh
