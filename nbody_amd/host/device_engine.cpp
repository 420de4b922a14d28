#include "device_engine.hpp"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <string>

namespace nbh {
namespace {
void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void nbpp_check(int rc) {
    if (rc != NBPP_OK) throw std::runtime_error(std::string("libnbpp: ") + nbpp_last_error());
}
void nbpm_check(int rc) {
    if (rc != NBPM_OK) throw std::runtime_error(std::string("libnbpm: ") + nbpm_last_error());
}
void nbgas_check(int rc) {
    if (rc != NBGAS_OK) throw std::runtime_error(std::string("libnbgas: ") + nbgas_last_error());
}
void nbmg_check(int rc) {
    if (rc != NBMG_OK) throw std::runtime_error(std::string("libnbmg: ") + nbmg_last_error());
}
}  // namespace

DeviceEngine::DeviceEngine(const Config& config, const ParticleSystem& s, const ZeldovichField* field)
    : config_(config) {
    const double* state[10] = {s.pos_x.data(), s.pos_y.data(), s.pos_z.data(), s.vel_x.data(), s.vel_y.data(),
                               s.vel_z.data(), s.acc_x.data(), s.acc_y.data(), s.acc_z.data(), s.mass.data()};
    construct(s.num_particles, state, field);
}

DeviceEngine::DeviceEngine(const Config& config, const DeviceParticles& s, const ZeldovichField* field)
    : config_(config) {
    // accelerations start at zero, as add_particle leaves them (src/particles.cpp:32-45)
    const double* state[10] = {s.x.data(), s.y.data(), s.z.data(), s.vx.data(), s.vy.data(),
                               s.vz.data(), nullptr, nullptr, nullptr, s.m.data()};
    construct(s.count, state, field);
}

// `state`: x y z vx vy vz ax ay az m of ALL particles, host or device memory (the copies find out
// which); a null entry stands for zeros.
void DeviceEngine::construct(std::size_t n_total, const double* const (&state)[10], const ZeldovichField* field) {
    n_ = n_total_ = n_total;
    ranks_ = nbmg_nranks();
    std::size_t padded = std::max<std::size_t>(n_total_, 1);
    if (ranks_ > 0) {
        nbmg_check(nbmg_partition(n_total_, nbmg_rank(), ranks_, &begin_, &n_));
        padded = std::max<std::size_t>(nbmg_padded_length(n_total_, ranks_), 1);
    }
    cells_ = static_cast<std::size_t>(config_.MESH_SIZE) * config_.MESH_SIZE * config_.MESH_SIZE;
    for (int c = 0; c < 9; ++c) {      // this process's block of the per-particle state
        hip_check(hipMalloc(reinterpret_cast<void**>(&dev_[c]), sizeof(double) * std::max<std::size_t>(n_, 1)), "hipMalloc");
        if (n_ && state[c]) hip_check(hipMemcpy(dev_[c], state[c] + begin_, sizeof(double) * n_, hipMemcpyDefault), "copy in");
        else if (n_) hip_check(hipMemset(dev_[c], 0, sizeof(double) * n_), "hipMemset");
    }
    // masses of ALL particles (the sources of the pair force), in a padded array
    hip_check(hipMalloc(reinterpret_cast<void**>(&dev_[9]), sizeof(double) * padded), "hipMalloc");
    hip_check(hipMemset(dev_[9], 0, sizeof(double) * padded), "hipMemset");
    if (n_total_) hip_check(hipMemcpy(dev_[9], state[9], sizeof(double) * n_total_, hipMemcpyDefault), "copy in");
    hip_check(hipDeviceSynchronize(), "sync");   // the fills ran on the null stream, the engine's stream does not wait for it
    if (ranks_ > 1) hip_check(hipMalloc(reinterpret_cast<void**>(&gather_), sizeof(double) * padded), "hipMalloc");
    int device = 0;
    hip_check(hipGetDevice(&device), "hipGetDevice");
    // One stream of the engine's own for everything: the three libraries would otherwise each fall
    // back to a default of their own (libnbpp to a private non-blocking stream, the others to the
    // null stream), with nothing ordering one library's kernels after another's.
    hip_check(hipStreamCreateWithFlags(reinterpret_cast<hipStream_t*>(&main_), hipStreamNonBlocking), "hipStreamCreate");
    for (void*& e : ev_) hip_check(hipEventCreate(reinterpret_cast<hipEvent_t*>(&e)), "hipEventCreate");
    // (not with several ranks: the collectives of one communicator must be issued in one order,
    // which two streams do not give)
    if (config_.USE_PM && config_.USE_PP && n_ && ranks_ <= 1) {
        // Its own HARDWARE queue: a process has four of them for all its streams of normal priority
        // (GPU_MAX_HW_QUEUES), handed out by use count, and a fifth stream anywhere in the process -
        // a library's private one, the runtime's own copy queue - can leave this one sharing the main
        // stream's queue, where the two run strictly one after the other (1.79 instead of 1.53 ms per
        // cycle at 128^3, seen once the initial conditions moved to the device).  Streams of another
        // priority have a pool of queues of their own.  Highest priority 1.53 ms per cycle, lowest 1.59.
        int least = 0, greatest = 0;
        hip_check(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
        hip_check(hipStreamCreateWithPriority(reinterpret_cast<hipStream_t*>(&pp_stream_), hipStreamNonBlocking, greatest),
                  "hipStreamCreateWithPriority");
        hip_check(hipEventCreateWithFlags(reinterpret_cast<hipEvent_t*>(&ev_pos_), hipEventDisableTiming), "hipEventCreate");
        hip_check(hipEventCreateWithFlags(reinterpret_cast<hipEvent_t*>(&ev_pp_), hipEventDisableTiming), "hipEventCreate");
        for (double*& a : pp_acc_) hip_check(hipMalloc(reinterpret_cast<void**>(&a), sizeof(double) * n_), "hipMalloc");
    }
    const bool meshes = config_.USE_PM || config_.USE_HYDRO;
    if (meshes) {
        nbpm_check(nbpm_init(device));
        nbpm_check(nbpm_set_timing(1));
        for (double*& m : mesh_) hip_check(hipMalloc(reinterpret_cast<void**>(&m), sizeof(double) * cells_), "hipMalloc");
    }
    cosmo.total_time = 0.0;
    update_cosmology(cosmo, config_);
    if (config_.USE_HYDRO) {                                                 // src/ics.cpp:241-288
        if (!field || field->dx.size() != cells_ || field->dy.size() != cells_ || field->dz.size() != cells_)
            throw std::runtime_error("USE_HYDRO needs the displacement field of the initial conditions");
        if (n_total_ == 0) throw std::runtime_error("USE_HYDRO needs dark-matter particles (src/ics.cpp:250)");
        nbgas_check(nbgas_init(device));
        for (double*& m : gas_) hip_check(hipMalloc(reinterpret_cast<void**>(&m), sizeof(double) * cells_), "hipMalloc");
        deposit_density();
        // the displacement field is only needed now: stage it in the gravity meshes
        const double* disp[3] = {field->dx.data(), field->dy.data(), field->dz.data()};
        for (int c = 0; c < 3; ++c)
            hip_check(hipMemcpyAsync(mesh_[2 + c], disp[c], sizeof(double) * cells_, hipMemcpyDeviceToDevice,
                                     static_cast<hipStream_t>(main_)), "stage displacement");
        const double total_dm_mass = n_total_ * config_.DM_PARTICLE_MASS;    // :251-252
        const double mass_ratio = config_.GAS_TOTAL_MASS / total_dm_mass;
        const double u0 = internal_energy_from_temperature(config_.INITIAL_GAS_TEMPERATURE_K, config_.HUBBLE_PARAM,
                                                           config_.BOX_SIZE_MPC, config_.DOMAIN_SIZE, config_.GAMMA);
        const nbgas_params gp = config_.gas_params();
        const nbgas_state gs = gas_state();
        nbgas_check(nbgas_initialize_device(&gp, &gs, mesh_[0], mesh_[2], mesh_[3], mesh_[4], mass_ratio,
                                            cosmo.hubble_param, field->f, config_.STANDING_PARTICLES ? 1 : 0, u0,
                                            main_));
    }
    compute_accelerations();
    current_dt = next_timestep();                                            // src/engine.cpp:16-22
}

DeviceEngine::~DeviceEngine() {
    (void)hipDeviceSynchronize();
    for (void* e : ev_)
        if (e) (void)hipEventDestroy(static_cast<hipEvent_t>(e));
    if (ev_pos_) (void)hipEventDestroy(static_cast<hipEvent_t>(ev_pos_));
    if (ev_pp_) (void)hipEventDestroy(static_cast<hipEvent_t>(ev_pp_));
    if (pp_stream_) (void)hipStreamDestroy(static_cast<hipStream_t>(pp_stream_));
    if (main_) (void)hipStreamDestroy(static_cast<hipStream_t>(main_));
    for (double* p : pp_acc_)
        if (p) (void)hipFree(p);
    for (double* p : dev_)
        if (p) (void)hipFree(p);
    if (gather_) (void)hipFree(gather_);
    if (phi_stage_) (void)hipFree(phi_stage_);
    for (double* p : gas_)
        if (p) (void)hipFree(p);
    for (double* p : mesh_)
        if (p) (void)hipFree(p);
}

nbgas_state DeviceEngine::gas_state() const {
    return nbgas_state{gas_[0], gas_[1], gas_[2], gas_[3], gas_[4], gas_[5], gas_[6], gas_[7], gas_[8]};
}

// bin_and_assign_mass's density (src/particles.cpp:47-100) of ALL particles into mesh_[0]: this
// process's block, then the sum over the processes.
void DeviceEngine::deposit_density() {
    const nbpm_params mp = config_.mesh_params();
    nbpm_check(nbpm_deposit_device(&mp, n_, dev_[0], dev_[1], dev_[2], dev_[9] + begin_, mesh_[0], main_));
    if (ranks_ > 1) nbmg_check(nbmg_allreduce_sum_device(cells_, mesh_[0], main_));
}

// The overlapped short-range force: queued on its own stream, into sums of its own, as soon as
// the positions of the cycle are final.  Its binning reads one count back, so the call blocks the
// host for a few tens of microseconds; the mesh work of the cycle (twenty launches) is therefore
// queued AFTER this call, while the pair kernel runs.  Queued the other way round, the pair
// kernel - the longest item of the cycle - started a third of a millisecond late at 128^3,
// behind the host's two dozen launches.  false: not on this cycle (timed, no mesh force,
// no second stream), the force phase then runs in order on the main stream.
bool DeviceEngine::launch_short_range() {
    const bool overlap = overlap_forces && !timed_ && pp_stream_ && pos_event_recorded_ && config_.USE_PM &&
                         config_.USE_PP && n_ && ranks_ <= 1;
    if (!overlap) return false;
    hipStream_t pps = static_cast<hipStream_t>(pp_stream_);
    const nbpp_params p = config_.pp_params();
    hip_check(hipStreamWaitEvent(pps, static_cast<hipEvent_t>(ev_pos_), 0), "wait");
    if (!pp_acc_zeroed_)
        for (double* a : pp_acc_) hip_check(hipMemsetAsync(a, 0, sizeof(double) * n_, pps), "zero");
    nbpp_check(nbpp_set_option(NBPP_OPT_TIMING, 0));
    // Four-wavefront workgroups while the mesh and gas kernels run beside the pair kernel: with one
    // wavefront per workgroup, the faster choice when it runs alone, the other stream's workgroups
    // hardly ever find room (1.69 against 1.57 ms per cycle at 128^3).  Once the particles have
    // clustered the pair kernel outlasts the mesh and gas work several times over and runs mostly
    // alone: one wavefront per workgroup from a clustering measure of 4 on (1 % over the first 16 000
    // cycles of the 128^3 run, which end at a = 0.55).  The same goes for the row walk with the pooled
    // drain, whose workgroups are single wavefronts: kept out of the lightly clustered cycles (1.58
    // against 1.52 ms per cycle over the first 100), the library's own choice afterwards.
    const bool alone = last_occupancy_ >= 4.0;
    nbpp_check(nbpp_set_option(NBPP_OPT_P3M_POOL_WORKGROUP, alone ? 64 : 256));
    nbpp_check(nbpp_set_option(NBPP_OPT_P3M_ROWS_POOLED, alone ? 0 : -1));   // its workgroups are single wavefronts too
    nbpp_check(nbpp_accumulate_device(&p, n_, dev_[0], dev_[1], dev_[2], dev_[9], 0, n_, pp_acc_[0], pp_acc_[1],
                                      pp_acc_[2], pp_stream_));
    nbpp_check(nbpp_set_option(NBPP_OPT_P3M_POOL_WORKGROUP, 0));
    nbpp_check(nbpp_set_option(NBPP_OPT_P3M_ROWS_POOLED, 0));
    nbpp_stats st;
    if (nbpp_get_stats(&st) == NBPP_OK) last_occupancy_ = st.cell_occupancy;
    hip_check(hipEventRecord(static_cast<hipEvent_t>(ev_pp_), pps), "event");
    return true;
}

void DeviceEngine::compute_accelerations() {
    hipStream_t ms = static_cast<hipStream_t>(main_);
    if (config_.USE_PM || config_.USE_HYDRO) {                               // :239-245
        // bin_and_assign_mass, total density, Poisson solve; the gas kick reads the meshes too
        const nbpm_params mp = config_.mesh_params();
        if (timed_) hip_check(hipEventRecord(static_cast<hipEvent_t>(ev_[0]), ms), "event");
        deposit_density();
        const double* source = mesh_[0];
        if (config_.USE_HYDRO) {                                             // :57-59
            const nbgas_params gp = config_.gas_params();
            const nbgas_state gs = gas_state();
            nbgas_check(nbgas_total_density_device(&gp, &gs, mesh_[0], mesh_[1], main_));
            source = mesh_[1];
        }
        nbpm_check(nbpm_solve_device(&mp, source, mesh_[2], mesh_[3], mesh_[4], main_));
        if (config_.USE_PM)
            nbpm_check(nbpm_interpolate_device(&mp, n_, dev_[0], dev_[1], dev_[2], mesh_[2], mesh_[3], mesh_[4],
                                               dev_[6], dev_[7], dev_[8], main_));
        if (timed_) {
            hip_check(hipEventRecord(static_cast<hipEvent_t>(ev_[1]), ms), "event");
            hip_check(hipEventSynchronize(static_cast<hipEvent_t>(ev_[1])), "event");
            float ms = 0.0f;
            hip_check(hipEventElapsedTime(&ms, static_cast<hipEvent_t>(ev_[0]), static_cast<hipEvent_t>(ev_[1])), "event");
            pm_ms += ms;
        }
        // The closing half kick of the gas (src/integrator.cpp:272-273) needs the mesh gradients
        // only: queued here it runs beside the short-range force instead of behind it.  The
        // particles' kick, which needs the sums, follows in step(); the two touch different
        // arrays, so their order does not matter.
        if (config_.USE_HYDRO && gas_kick_after_mesh_ > 0.0) {
            const nbgas_params gp = config_.gas_params();
            const nbgas_state gs = gas_state();
            // ... and nothing touches the gas until the next cycle: the CFL limit of the next step
            // (src/engine.cpp:18) is reduced in the same pass and read back now, collected in
            // cfl_timestep()
            if (config_.USE_ADAPTIVE_DT) {
                nbgas_check(nbgas_kick_cfl_enqueue_device(&gp, &gs, mesh_[2], mesh_[3], mesh_[4], gas_kick_after_mesh_,
                                                          cosmo.scale_factor, cosmo.hubble_param, main_));
                cfl_in_flight_ = true;
            } else {
                nbgas_check(nbgas_kick_device(&gp, &gs, mesh_[2], mesh_[3], mesh_[4], gas_kick_after_mesh_,
                                              cosmo.scale_factor, cosmo.hubble_param, main_));
            }
            gas_kick_after_mesh_ = 0.0;
        }
    }
    if (!config_.USE_PM) {                                                   // :246-251
        for (int c = 6; c < 9; ++c)
            hip_check(hipMemsetAsync(dev_[c], 0, sizeof(double) * n_, ms), "zero acc");
        hip_check(hipStreamSynchronize(ms), "sync");
    }
    if (pp_launched_ || launch_short_range()) {
        // the short-range force has been running on its own stream since the drift (step()) or has
        // been started just now (initial force phase); the main stream waits for its sums
        pp_launched_ = false;
        hip_check(hipStreamWaitEvent(ms, static_cast<hipEvent_t>(ev_pp_), 0), "wait");
        // added and cleared in one pass: the next evaluation (which waits for this cycle's drift,
        // queued behind this kernel) finds zeros without a pass of its own in front of it
        if (kick_after_forces_ > 0.0 && ranks_ == 0) {
            // ... the particles' closing half kick (:274) and max |a|^2 (:261-262) in the same pass
            nbpp_check(nbpp_add_clear_kick_max_device(n_, dev_[6], dev_[7], dev_[8], pp_acc_[0], pp_acc_[1],
                                                      pp_acc_[2], dev_[3], dev_[4], dev_[5], kick_after_forces_,
                                                      cosmo.scale_factor, cosmo.hubble_param, &max_accel_sq, main_));
            kick_fused_ = true;
            max_fused_ = true;
        } else if (kick_after_forces_ > 0.0) {   // ... and the particles' closing half kick (:274) in the same pass
            nbpp_check(nbpp_add_clear_kick_device(n_, dev_[6], dev_[7], dev_[8], pp_acc_[0], pp_acc_[1], pp_acc_[2],
                                                  dev_[3], dev_[4], dev_[5], kick_after_forces_, cosmo.scale_factor,
                                                  cosmo.hubble_param, main_));
            kick_fused_ = true;
        } else {
            nbpp_check(nbpp_add_clear_device(n_, dev_[6], dev_[7], dev_[8], pp_acc_[0], pp_acc_[1], pp_acc_[2], main_));
        }
        pp_acc_zeroed_ = true;
    } else if (config_.USE_PP && n_total_ && ranks_ > 0) {                   // :254-258, over the ranks
        const nbpp_params p = config_.pp_params();
        nbpp_check(nbpp_set_option(NBPP_OPT_TIMING, 0));
        nbmg_check(nbmg_set_option(NBMG_OPT_TIMING, timed_ ? 1 : 0));
        nbmg_check(nbmg_accumulate_device(&p, n_total_, dev_[0], dev_[1], dev_[2], dev_[9], dev_[6], dev_[7],
                                          dev_[8], main_));
        nbmg_stats st;
        if (timed_ && nbmg_get_stats(&st) == NBMG_OK) {
            pp_ms += st.total_ms;
            comm_ms += st.allgather_ms + st.reduce_ms;
        }
    } else if (config_.USE_PP && n_) {                                       // :254-258
        const nbpp_params p = config_.pp_params();
        nbpp_check(nbpp_set_option(NBPP_OPT_TIMING, timed_ ? 1 : 0));
        nbpp_check(nbpp_accumulate_device(&p, n_, dev_[0], dev_[1], dev_[2], dev_[9], 0, n_,
                                          dev_[6], dev_[7], dev_[8], main_));
        nbpp_stats st;
        if (timed_ && nbpp_get_stats(&st) == NBPP_OK) pp_ms += st.total_ms;
    }
    pos_event_recorded_ = false;
    if (timed_) ++timed_force_calls;
    if (max_fused_) {
        max_fused_ = false;                                                  // read back with the sum and the kick
    } else if (!config_.STANDING_PARTICLES && ranks_ > 0) {                  // :261-262, over the ranks
        nbmg_check(nbmg_max_accel_sq_device(n_, dev_[6], dev_[7], dev_[8], &max_accel_sq, main_));
    } else if (!config_.STANDING_PARTICLES) {
        nbpp_check(nbpp_max_accel_sq_device(n_, dev_[6], dev_[7], dev_[8], &max_accel_sq, main_));
    } else {                                                                 // :263-269
        nbpp_check(nbpp_max_accel_sq_device(0, nullptr, nullptr, nullptr, &max_accel_sq, main_));
        for (int c = 6; c < 9; ++c)
            hip_check(hipMemsetAsync(dev_[c], 0, sizeof(double) * n_, ms), "zero acc");
        hip_check(hipStreamSynchronize(ms), "sync");
        max_accel_sq = 1e-9;
    }
}

double DeviceEngine::gravity_timestep() const {
    if (n_total_ == 0) return std::numeric_limits<double>::infinity();
    return std::sqrt(config_.SOFTENING_SQUARED) / std::sqrt(max_accel_sq) * config_.GRAVITY_DT_FACTOR;
}

double DeviceEngine::cfl_timestep() {
    double dt_cfl = std::numeric_limits<double>::infinity();                 // src/gas.cpp:302
    if (config_.USE_HYDRO) {
        const nbgas_params gp = config_.gas_params();
        const nbgas_state gs = gas_state();
        if (cfl_in_flight_) nbgas_check(nbgas_cfl_timestep_finish(&gp, &dt_cfl));
        else nbgas_check(nbgas_cfl_timestep_device(&gp, &gs, &dt_cfl, main_));
        cfl_in_flight_ = false;
    }
    return dt_cfl;
}

double DeviceEngine::next_timestep() {
    if (!config_.USE_ADAPTIVE_DT) return config_.FIXED_DT;
    return std::min({cfl_timestep(), gravity_timestep(), config_.FIXED_DT});
}

void DeviceEngine::download_mesh_solve(std::vector<double>& total_rho, std::vector<double>& phi) {
    total_rho.clear();
    phi.clear();
    if (cells_ == 0) return;
    hip_check(hipDeviceSynchronize(), "sync");
    const nbpm_params mp = config_.mesh_params();
    if (!mesh_[0]) {
        // Pair force only, no gas: the step itself needs no mesh, but the reference's KDK_step
        // assigns the density and solves for the potential on every cycle whatever the switches
        // (src/integrator.cpp:239-240), so its diagnostics always have them.  Here they are
        // computed when asked for, from the positions the particles have now.
        int device = 0;
        hip_check(hipGetDevice(&device), "hipGetDevice");
        nbpm_check(nbpm_init(device));
        for (double*& m : mesh_) hip_check(hipMalloc(reinterpret_cast<void**>(&m), sizeof(double) * cells_), "hipMalloc");
        meshes_on_demand_ = true;
    }
    if (meshes_on_demand_) {
        deposit_density();
        nbpm_check(nbpm_solve_device(&mp, mesh_[0], mesh_[2], mesh_[3], mesh_[4], main_));
    }
    total_rho.resize(cells_);
    phi.resize(cells_);
    const double* source = config_.USE_HYDRO ? mesh_[1] : mesh_[0];
    hip_check(hipStreamSynchronize(static_cast<hipStream_t>(main_)), "sync");
    hip_check(hipMemcpy(total_rho.data(), source, sizeof(double) * cells_, hipMemcpyDeviceToHost), "download");
    // the potential comes through a buffer of its own: the density meshes stay what they are, so
    // a second call before the next force phase returns the same two fields
    if (!phi_stage_) hip_check(hipMalloc(reinterpret_cast<void**>(&phi_stage_), sizeof(double) * cells_), "hipMalloc");
    nbpm_check(nbpm_potential_device(&mp, phi_stage_, main_));
    hip_check(hipStreamSynchronize(static_cast<hipStream_t>(main_)), "sync");
    hip_check(hipMemcpy(phi.data(), phi_stage_, sizeof(double) * cells_, hipMemcpyDeviceToHost), "download");
}

void DeviceEngine::step() {
    hipStream_t ms = static_cast<hipStream_t>(main_);
    timed_ = timing_every > 0 && (cycle_count + 1) % timing_every == 0;
    const double dt = current_dt;
    const nbgas_params gp = config_.gas_params();
    const nbgas_state gs = gas_state();
    update_cosmology(cosmo, config_);                                                   // integrator.cpp:216
    // The particles first: the short-range force of this cycle starts on its own stream as soon as
    // they have drifted.  The gas's opening half kick (:219-220) touches other arrays, so it may
    // follow the drift instead of preceding the particles' kick.
    nbpp_check(nbpp_kick_drift_device(n_, dev_[0], dev_[1], dev_[2], dev_[3], dev_[4], dev_[5], dev_[6], dev_[7],
                                      dev_[8], dt / 2.0, dt, cosmo.scale_factor, cosmo.hubble_param,
                                      config_.DOMAIN_SIZE, main_));                   // :221 and :224 in one pass
    const bool pp_early = pp_stream_ && overlap_forces && !timed_;
    if (pp_early) {
        hip_check(hipEventRecord(static_cast<hipEvent_t>(ev_pos_), ms), "event");
        pos_event_recorded_ = true;
    }
    if (config_.USE_HYDRO)                                                              // :219-220
        nbgas_check(nbgas_kick_device(&gp, &gs, mesh_[2], mesh_[3], mesh_[4], dt / 2.0, cosmo.scale_factor,
                                      cosmo.hubble_param, main_));
    if (config_.USE_HYDRO) {                                                            // :227-230
        if (timed_) hip_check(hipEventRecord(static_cast<hipEvent_t>(ev_[2]), ms), "event");
        nbgas_check(nbgas_hydro_step_device(&gp, &gs, dt, main_));
        if (timed_) hip_check(hipEventRecord(static_cast<hipEvent_t>(ev_[3]), ms), "event");
    }
    // The short-range call comes behind the gas's five launches and before the mesh's twenty: the
    // sweeps then run beside the short-range binning, which leaves most of the GPU idle, instead of
    // beside the pair kernel (1.52-1.53 against 1.54-1.55 ms per cycle at 128^3).
    if (pp_early) pp_launched_ = launch_short_range();
    cosmo.total_time += dt;                                                             // :233
    update_cosmology(cosmo, config_);                                                   // :234
    gas_kick_after_mesh_ = config_.USE_HYDRO ? dt / 2.0 : 0.0;                          // :272-273, see there
    // :274, in one pass with the sum of mesh and short-range force where the cycle has one (standing
    // particles are kicked with the zeroed accelerations, after compute_accelerations() has cleared them)
    kick_after_forces_ = config_.STANDING_PARTICLES ? 0.0 : dt / 2.0;
    kick_fused_ = false;
    compute_accelerations();                                                            // :236-269
    kick_after_forces_ = 0.0;
    if (!kick_fused_)
        nbpp_check(nbpp_kick_device(n_, dev_[3], dev_[4], dev_[5], dev_[6], dev_[7], dev_[8], dt / 2.0,
                                    cosmo.scale_factor, cosmo.hubble_param, main_));  // :274
    ++cycle_count;
    if (timed_) ++timed_steps;
    if (config_.USE_HYDRO && timed_) {
        hip_check(hipEventSynchronize(static_cast<hipEvent_t>(ev_[3])), "event");
        float ms = 0.0f;
        hip_check(hipEventElapsedTime(&ms, static_cast<hipEvent_t>(ev_[2]), static_cast<hipEvent_t>(ev_[3])), "event");
        gas_ms += ms;
    }
    if (config_.USE_ADAPTIVE_DT) current_dt = next_timestep();                          // engine.cpp:47-51
    needs_more_cycles = cosmo.scale_factor < config_.MAX_SCALE_FACTOR &&
                        cycle_count < config_.MAX_CYCLES;                               // :53-54
}

ExitStatus DeviceEngine::run() {
    while (!stop_requested_ && needs_more_cycles) step();
    if (stop_requested_) return ExitStatus::UserAborted;
    if (cycle_count >= config_.MAX_CYCLES) return ExitStatus::ReachedMaxCycles;
    return ExitStatus::ReachedMaxScaleFactor;
}

void DeviceEngine::download(ParticleSystem& out) {
    std::vector<double>* host[10] = {&out.pos_x, &out.pos_y, &out.pos_z, &out.vel_x, &out.vel_y,
                                     &out.vel_z, &out.acc_x, &out.acc_y, &out.acc_z, &out.mass};
    hip_check(hipDeviceSynchronize(), "sync");
    for (int c = 0; c < 10; ++c) {
        host[c]->resize(n_total_);
        if (!n_total_) continue;
        const double* from = dev_[c];
        if (ranks_ > 1 && c < 9) {      // every rank ends up with the whole state (collective)
            nbmg_check(nbmg_allgather_device(n_total_, dev_[c], gather_, main_));
            hip_check(hipStreamSynchronize(static_cast<hipStream_t>(main_)), "sync");
            from = gather_;
        }
        hip_check(hipMemcpy(host[c]->data(), from, sizeof(double) * n_total_, hipMemcpyDeviceToHost), "download");
    }
    out.num_particles = n_total_;
    out.max_accel_sq = max_accel_sq;
}

void DeviceEngine::download_gas(std::vector<double> (&out)[9]) {
    hip_check(hipDeviceSynchronize(), "sync");
    for (int c = 0; c < 9; ++c) {
        out[c].resize(gas_[c] ? cells_ : 0);
        if (gas_[c]) hip_check(hipMemcpy(out[c].data(), gas_[c], sizeof(double) * cells_, hipMemcpyDeviceToHost), "download");
    }
}

}  // namespace nbh
