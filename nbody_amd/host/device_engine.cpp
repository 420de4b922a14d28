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
}  // namespace

DeviceEngine::DeviceEngine(const Config& config, const ParticleSystem& s) : config_(config) {
    n_ = s.num_particles;
    const std::vector<double>* host[10] = {&s.pos_x, &s.pos_y, &s.pos_z, &s.vel_x, &s.vel_y,
                                           &s.vel_z, &s.acc_x, &s.acc_y, &s.acc_z, &s.mass};
    for (int c = 0; c < 10; ++c) {
        hip_check(hipMalloc(reinterpret_cast<void**>(&dev_[c]), sizeof(double) * std::max<std::size_t>(n_, 1)), "hipMalloc");
        if (n_) hip_check(hipMemcpy(dev_[c], host[c]->data(), sizeof(double) * n_, hipMemcpyHostToDevice), "upload");
    }
    if (config_.USE_PM) {
        int device = 0;
        hip_check(hipGetDevice(&device), "hipGetDevice");
        nbpm_check(nbpm_init(device));
        nbpm_check(nbpm_set_timing(1));
    }
    cosmo.total_time = 0.0;
    update_cosmology(cosmo, config_);
    compute_accelerations();
    current_dt = config_.USE_ADAPTIVE_DT ? std::min(gravity_timestep(), config_.FIXED_DT)
                                         : config_.FIXED_DT;               // engine.cpp:18-22
}

DeviceEngine::~DeviceEngine() {
    for (double* p : dev_)
        if (p) (void)hipFree(p);
}

void DeviceEngine::compute_accelerations() {
    if (config_.USE_PM) {                                                               // :239-245
        const nbpm_params mp = config_.mesh_params();
        nbpm_check(nbpm_accelerations_device(&mp, n_, dev_[0], dev_[1], dev_[2], dev_[9], dev_[6],
                                             dev_[7], dev_[8], nullptr));
        nbpm_stats ms;
        if (nbpm_get_stats(&ms) == NBPM_OK) pm_ms += ms.total_ms;
    } else {
        for (int c = 6; c < 9; ++c)
            hip_check(hipMemsetAsync(dev_[c], 0, sizeof(double) * n_, nullptr), "zero acc");  // :246-251
        hip_check(hipStreamSynchronize(nullptr), "sync");
    }
    if (config_.USE_PP && n_) {                                                         // :254-258
        const nbpp_params p = config_.pp_params();
        nbpp_check(nbpp_accumulate_device(&p, n_, dev_[0], dev_[1], dev_[2], dev_[9], 0, n_,
                                          dev_[6], dev_[7], dev_[8], nullptr));
        nbpp_stats st;
        if (nbpp_get_stats(&st) == NBPP_OK) pp_ms += st.total_ms;
    }
    if (!config_.STANDING_PARTICLES) {                                                  // :261-262
        nbpp_check(nbpp_max_accel_sq_device(n_, dev_[6], dev_[7], dev_[8], &max_accel_sq, nullptr));
    } else {                                                                            // :263-269
        nbpp_check(nbpp_max_accel_sq_device(0, nullptr, nullptr, nullptr, &max_accel_sq, nullptr));
        for (int c = 6; c < 9; ++c)
            hip_check(hipMemsetAsync(dev_[c], 0, sizeof(double) * n_, nullptr), "zero acc");
        hip_check(hipStreamSynchronize(nullptr), "sync");
        max_accel_sq = 1e-9;
    }
}

double DeviceEngine::gravity_timestep() const {
    if (n_ == 0) return std::numeric_limits<double>::infinity();
    return std::sqrt(config_.SOFTENING_SQUARED) / std::sqrt(max_accel_sq) * config_.GRAVITY_DT_FACTOR;
}

void DeviceEngine::step() {
    const double dt = current_dt;
    update_cosmology(cosmo, config_);                                                   // integrator.cpp:216
    nbpp_check(nbpp_kick_device(n_, dev_[3], dev_[4], dev_[5], dev_[6], dev_[7], dev_[8], dt / 2.0,
                                cosmo.scale_factor, cosmo.hubble_param, nullptr));      // :221
    nbpp_check(nbpp_drift_device(n_, dev_[0], dev_[1], dev_[2], dev_[3], dev_[4], dev_[5], dt,
                                 config_.DOMAIN_SIZE, nullptr));                        // :224
    cosmo.total_time += dt;                                                             // :233
    update_cosmology(cosmo, config_);                                                   // :234
    compute_accelerations();                                                            // :236-269
    nbpp_check(nbpp_kick_device(n_, dev_[3], dev_[4], dev_[5], dev_[6], dev_[7], dev_[8], dt / 2.0,
                                cosmo.scale_factor, cosmo.hubble_param, nullptr));      // :274
    ++cycle_count;
    if (config_.USE_ADAPTIVE_DT)                                                        // engine.cpp:47-51
        current_dt = std::min(gravity_timestep(), config_.FIXED_DT);
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
        host[c]->resize(n_);
        if (n_) hip_check(hipMemcpy(host[c]->data(), dev_[c], sizeof(double) * n_, hipMemcpyDeviceToHost), "download");
    }
    out.num_particles = n_;
    out.max_accel_sq = max_accel_sq;
}

}  // namespace nbh
