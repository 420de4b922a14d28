#include "integrator.hpp"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <initializer_list>
#include <stdexcept>
#include <string>

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

// ---- kick, drift and max |a|^2 of a HOST-resident ParticleSystem ---------------------------------
// The arithmetic of src/integrator.cpp:168-213 exists once in this code base, as device kernels
// behind nbpp_kick_device / nbpp_kick_drift_device / nbpp_max_accel_sq_device.  A caller that keeps
// its state in host vectors, as the reference does, gets the same kernels through a device mirror
// of the arrays a call touches: up, one kernel, down.  (The engine proper never takes this path: its
// state lives in HBM, DeviceEngine.)
namespace {

void hip_ok(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void nbpp_ok(int rc) {
    if (rc != NBPP_OK) throw std::runtime_error(std::string("libnbpp: ") + nbpp_last_error());
}

class Mirror {   // device copies of position, velocity and acceleration, grown on demand
   public:
    ~Mirror() {
        for (double* p : dev_)
            if (p) (void)hipFree(p);
        if (stream_) (void)hipStreamDestroy(stream_);
    }
    void* stream() {   // copies and kernels of one call are ordered on one stream of the mirror's own
        if (!stream_) hip_ok(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
        return stream_;
    }
    enum Field { X = 0, V = 3, A = 6 };
    double* operator[](int c) const { return dev_[c]; }
    void up(std::size_t n, std::vector<double>* const (&host)[9], std::initializer_list<Field> fields) {
        if (n > cap_) {
            for (double*& p : dev_) {
                if (p) hip_ok(hipFree(p), "hipFree");
                p = nullptr;
                hip_ok(hipMalloc(reinterpret_cast<void**>(&p), sizeof(double) * n), "hipMalloc");
            }
            cap_ = n;
        }
        for (Field f : fields)
            for (int c = f; c < f + 3; ++c)
                hip_ok(hipMemcpyAsync(dev_[c], host[c]->data(), sizeof(double) * n, hipMemcpyHostToDevice,
                                      static_cast<hipStream_t>(stream())), "upload");
    }
    void down(std::size_t n, std::vector<double>* const (&host)[9], std::initializer_list<Field> fields) {
        for (Field f : fields)
            for (int c = f; c < f + 3; ++c)
                hip_ok(hipMemcpyAsync(host[c]->data(), dev_[c], sizeof(double) * n, hipMemcpyDeviceToHost,
                                      static_cast<hipStream_t>(stream())), "download");
        hip_ok(hipStreamSynchronize(static_cast<hipStream_t>(stream())), "sync");
    }

   private:
    double* dev_[9] = {};
    std::size_t cap_ = 0;
    hipStream_t stream_ = nullptr;
};

Mirror& mirror() {
    static Mirror m;
    return m;
}

}  // namespace

#define STATE_OF(dm)                                                                                         \
    std::vector<double>* const state[9] = {&(dm).pos_x, &(dm).pos_y, &(dm).pos_z, &(dm).vel_x, &(dm).vel_y, \
                                           &(dm).vel_z, &(dm).acc_x, &(dm).acc_y, &(dm).acc_z}

void apply_dm_kick(ParticleSystem& dm, double dt, double a, double H) {
    const std::size_t n = dm.num_particles;
    if (!n) return;
    STATE_OF(dm);
    Mirror& d = mirror();
    d.up(n, state, {Mirror::V, Mirror::A});
    nbpp_ok(nbpp_kick_device(n, d[3], d[4], d[5], d[6], d[7], d[8], dt, a, H, d.stream()));
    d.down(n, state, {Mirror::V});
}

void apply_dm_drift(ParticleSystem& dm, double dt, double domain_size) {
    const std::size_t n = dm.num_particles;
    if (!n) return;
    STATE_OF(dm);
    Mirror& d = mirror();
    d.up(n, state, {Mirror::X, Mirror::V});
    nbpp_ok(nbpp_drift_device(n, d[0], d[1], d[2], d[3], d[4], d[5], dt, domain_size, d.stream()));
    d.down(n, state, {Mirror::X});
}

void calculate_max_acceleration(ParticleSystem& dm) {
    const std::size_t n = dm.num_particles;
    STATE_OF(dm);
    Mirror& d = mirror();
    if (n) d.up(n, state, {Mirror::A});
    // (no particles: the floor, 1e-9)
    nbpp_ok(nbpp_max_accel_sq_device(n, d[6], d[7], d[8], &dm.max_accel_sq, d.stream()));
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
