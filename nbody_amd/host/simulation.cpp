// simulation.cpp - see simulation.hpp.
#include "simulation.hpp"

#include <algorithm>
#include <cstdio>
#include <filesystem>
#include <stdexcept>

#include "particles.hpp"

namespace nbh {

static bool writer_rank();

Simulation::Simulation(const Config& config, const RunOptions& options) : config_(config), options_(options) {
    Cosmology cosmo;
    ZeldovichField field;
    // field and particles are generated on the device and stay there
    const DeviceParticles dm = initialize_particles_device(config_, cosmo, &field);
    engine_.reset(new DeviceEngine(config_, dm, config_.USE_HYDRO ? &field : nullptr));
    engine_->overlap_forces = options_.overlap_forces;
    // device times are for the log: none are taken when nothing is logged
    engine_->timing_every = options_.timing_every >= 0 ? options_.timing_every
                            : config_.DEBUG_INFO_EVERY_CYCLES <= 0
                                ? 0
                                : std::max(1, std::min(config_.DEBUG_INFO_EVERY_CYCLES, config_.MAX_CYCLES));
    if (writer_rank() && !options_.snapshot_target.empty() && std::filesystem::is_directory(options_.snapshot_target))
        hdf5_.reset(new SnapshotWriter(options_.snapshot_target, config_));
    if (writer_rank() && !options_.log_dir.empty()) logger_.reset(new Logger(options_.log_dir));
    next_output_a_ = config_.START_A + config_.SAVE_HDF5_EVERY_DELTA_A;
    logged_wall_ = std::chrono::steady_clock::now();
}

// With several processes (libnbmg) every one of them downloads, since that is a collective
// operation, and the first alone prints and writes.
static bool writer_rank() { return nbmg_rank() <= 0; }

void Simulation::report(const char* when) {
    DeviceEngine& engine = *engine_;
    ParticleSystem now(config_);
    engine.download(now);
    if (!writer_rank()) return;
    std::printf("%s: cycle %ld a=%.17g dt=%.17g max|a|^2=%.17g\n", when, engine.cycle_count,
                engine.cosmo.scale_factor, engine.current_dt, engine.max_accel_sq);
    if (now.num_particles)
        std::printf("%s: p0 acc %.17g %.17g %.17g\n", when, now.acc_x[0], now.acc_y[0], now.acc_z[0]);
    std::vector<double> gas[9];
    engine.download_gas(gas);
    if (!gas[0].empty()) {
        double mass = 0.0, energy = 0.0;
        for (double v : gas[0]) mass += v;
        for (double v : gas[4]) energy += v;
        std::printf("%s: gas mass %.17g energy %.17g\n", when, mass * config_.CELL_VOLUME, energy * config_.CELL_VOLUME);
    }
}

void Simulation::save_snapshot() {
    if (options_.snapshot_target.empty()) return;
    DeviceEngine& engine = *engine_;
    ParticleSystem now(config_);
    engine.download(now);
    std::vector<double> gas[9];
    engine.download_gas(gas);
    if (!writer_rank()) {
        ++snapshot_count_;
        return;
    }
    if (hdf5_) {
        hdf5_->save_snapshot(snapshot_count_++, engine.cycle_count, now, gas, engine.cosmo.total_time,
                             engine.cosmo.scale_factor, config_);
        return;
    }
    char index[32];
    std::snprintf(index, sizeof(index), "_%04d.bin", snapshot_count_++);
    const std::string name = options_.snapshot_target + index;
    FILE* f = std::fopen(name.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot open snapshot file " + name);
    const double head[6] = {static_cast<double>(now.num_particles), static_cast<double>(config_.MESH_SIZE),
                            engine.cosmo.scale_factor, engine.cosmo.total_time, config_.DOMAIN_SIZE,
                            config_.USE_HYDRO ? 1.0 : 0.0};
    std::fwrite(head, sizeof(double), 6, f);
    for (auto* v : {&now.pos_x, &now.pos_y, &now.pos_z, &now.mass}) std::fwrite(v->data(), sizeof(double), v->size(), f);
    if (config_.USE_HYDRO) {
        std::fwrite(gas[0].data(), sizeof(double), gas[0].size(), f);
        std::fwrite(gas[4].data(), sizeof(double), gas[4].size(), f);
    }
    std::fclose(f);
}

void Simulation::log_diagnostics() {
    if (options_.log_dir.empty()) return;
    DeviceEngine& engine = *engine_;
    ParticleSystem now(config_);
    // Collective with several processes (libnbmg): download() all-gathers the particles, and
    // download_mesh_solve() all-reduces the density when the step itself keeps no mesh (pair force
    // only, no gas).  Every process therefore makes both calls, in this order, before the ones
    // that do not write leave; a process that left earlier would meet the others' all-reduce with
    // the next cycle's all-gather.
    engine.download(now);
    std::vector<double> gas[9], total_rho, phi;
    engine.download_mesh_solve(total_rho, phi);
    if (!logger_ || !logger_->is_open()) return;
    engine.download_gas(gas);
    Diagnostics d = compute_diagnostics(config_, now, gas, total_rho, phi, engine.cosmo, engine.cfl_timestep(),
                                        engine.gravity_timestep(), engine.current_dt);
    d.cycle = engine.cycle_count;
    const long cycles = engine.cycle_count - logged_cycle_;
    const auto wall_now = std::chrono::steady_clock::now();
    if (cycles > 0) {   // averages since the last line
        d.avg_step = std::chrono::duration<double>(wall_now - logged_wall_).count() / cycles;
        // device times: averages over the cycles since the last line that were timed (none timed,
        // as can happen for the last line of a run: the averages before it)
        const long calls = engine.timed_force_calls - logged_calls_;
        const long steps = engine.timed_steps - logged_steps_;
        if (calls > 0) {
            last_avg_[0] = (engine.pm_ms - logged_ms_[0]) * 1e-3 / calls;
            last_avg_[1] = (engine.pp_ms - logged_ms_[1]) * 1e-3 / calls;
        }
        if (steps > 0) last_avg_[2] = (engine.gas_ms - logged_ms_[2]) * 1e-3 / steps;
        d.avg_pm = last_avg_[0];
        d.avg_pp = last_avg_[1];
        d.avg_hydro = last_avg_[2];
    }
    logger_->log(d);
    logged_cycle_ = engine.cycle_count;
    logged_ms_[0] = engine.pm_ms; logged_ms_[1] = engine.pp_ms; logged_ms_[2] = engine.gas_ms;
    logged_calls_ = engine.timed_force_calls; logged_steps_ = engine.timed_steps;
    logged_wall_ = std::chrono::steady_clock::now();
}

ExitStatus Simulation::run(const volatile std::sig_atomic_t* stop) {
    DeviceEngine& engine = *engine_;
    // the first force phase also paid for set-up (FFT plans, buffers): keep it out of the averages
    engine.pp_ms = engine.pm_ms = engine.gas_ms = 0.0;
    engine.timed_force_calls = engine.timed_steps = 0;
    if (config_.SAVE_HDF5_EVERY_DELTA_A > 0.0) save_snapshot();
    log_diagnostics();
    const auto t0 = std::chrono::steady_clock::now();
    auto stopped = [&]() { return stop && *stop; };
    // DeviceEngine::run() (src/engine.cpp:78-91) unrolled, to log and save like the reference
    // does inside its step (src/engine.cpp:56-75)
    while (engine.needs_more_cycles && !stopped()) {
        engine.step();
        const bool reporting = config_.DEBUG_INFO_EVERY_CYCLES > 0 && engine.cycle_count % config_.DEBUG_INFO_EVERY_CYCLES == 0;
        if (reporting && options_.progress) {
            nbpp_stats ps{};
            nbpp_get_stats(&ps);
            std::printf("cycle %ld: a=%.6f dt=%.4e max|a|^2=%.4e | last PP call %.3f ms (binning %.3f, pairs %.3f), %.4g accepted pairs (%.1f G/s)\n",
                        engine.cycle_count, engine.cosmo.scale_factor, engine.current_dt, engine.max_accel_sq,
                        ps.total_ms, ps.prepare_ms, ps.pair_kernel_ms, static_cast<double>(ps.pairs_accepted),
                        ps.total_ms > 0 ? ps.pairs_accepted / ps.total_ms * 1e-6 : 0.0);
            std::fflush(stdout);
        }
        const bool last_cycle = !engine.needs_more_cycles || stopped();
        if (config_.DEBUG_INFO_EVERY_CYCLES > 0 && (last_cycle || reporting)) log_diagnostics();
        const bool must_save = engine.cosmo.scale_factor >= next_output_a_;
        if (config_.SAVE_HDF5_EVERY_DELTA_A > 0.0 && (last_cycle || must_save)) {
            save_snapshot();
            while (next_output_a_ <= engine.cosmo.scale_factor) next_output_a_ += config_.SAVE_HDF5_EVERY_DELTA_A;
        }
    }
    run_wall_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stopped()) return ExitStatus::UserAborted;
    return engine.cycle_count >= config_.MAX_CYCLES ? ExitStatus::ReachedMaxCycles : ExitStatus::ReachedMaxScaleFactor;
}

void Simulation::print_timing(ExitStatus status) const {
    const DeviceEngine& engine = *engine_;
    if (!writer_rank()) return;
    if (nbmg_nranks() > 1)
        std::printf("ranks: %d processes, one GPU each; of the PP time %.3f ms/cycle is all-gather and reduce-scatter\n",
                    nbmg_nranks(), engine.comm_ms / std::max(1L, engine.timed_force_calls));
    std::printf("timing: %ld cycles, %.3f ms/cycle wall; PP %.3f, mesh %.3f, hydro sweeps %.3f ms/cycle (device, %ld timed cycles); exit=%d\n",
                engine.cycle_count, run_wall_ms_ / std::max(1L, engine.cycle_count),
                engine.pp_ms / std::max(1L, engine.timed_force_calls), engine.pm_ms / std::max(1L, engine.timed_force_calls),
                engine.gas_ms / std::max(1L, engine.timed_steps), engine.timed_steps, static_cast<int>(status));
}

}  // namespace nbh
