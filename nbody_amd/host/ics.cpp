#include "ics.hpp"

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>

namespace nbh {
namespace {

void hip_ok(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void nbpm_ok(int rc) {
    if (rc != NBPM_OK) throw std::runtime_error(std::string("libnbpm: ") + nbpm_last_error());
}

// libnbpm on the device this process already uses (the current one)
void mesh_library_ready() {
    int device = 0;
    hip_ok(hipGetDevice(&device), "hipGetDevice");
    nbpm_ok(nbpm_init(device));
}

nbpm_ic_params ic_params(const Config& config, double scale_factor) {
    nbpm_ic_params p{};
    p.mesh_size = config.MESH_SIZE;
    p.n_per_side = config.N_PER_SIDE;
    p.domain_size = config.DOMAIN_SIZE;
    p.box_size_mpc = config.BOX_SIZE_MPC;
    p.hubble_param = config.HUBBLE_PARAM;
    p.omega_m = config.OMEGA_M;
    p.spectral_index = config.SPECTRAL_INDEX;
    p.scale_factor = scale_factor;
    nbpm_ok(nbpm_sigma8_amplitude(config.SIGMA_8, config.OMEGA_M, config.HUBBLE_PARAM, config.SPECTRAL_INDEX,
                                  &p.amplitude));
    return p;
}

}  // namespace

DeviceArray::DeviceArray(std::size_t count) : count_(count) {
    if (count) hip_ok(hipMalloc(reinterpret_cast<void**>(&data_), sizeof(double) * count), "hipMalloc");
}
DeviceArray::~DeviceArray() {
    if (data_) (void)hipFree(data_);
}
DeviceArray::DeviceArray(DeviceArray&& other) noexcept : data_(other.data_), count_(other.count_) {
    other.data_ = nullptr;
    other.count_ = 0;
}
DeviceArray& DeviceArray::operator=(DeviceArray&& other) noexcept {
    if (this != &other) {
        if (data_) (void)hipFree(data_);
        data_ = std::exchange(other.data_, nullptr);
        count_ = std::exchange(other.count_, 0);
    }
    return *this;
}
void DeviceArray::upload(const std::vector<double>& host) {
    if (host.size() != count_) *this = DeviceArray(host.size());
    if (count_) hip_ok(hipMemcpy(data_, host.data(), sizeof(double) * count_, hipMemcpyHostToDevice), "upload");
}
std::vector<double> DeviceArray::download() const {
    std::vector<double> host(count_);
    hip_ok(hipDeviceSynchronize(), "sync");
    if (count_) hip_ok(hipMemcpy(host.data(), data_, sizeof(double) * count_, hipMemcpyDeviceToHost), "download");
    return host;
}

ZeldovichField compute_zeldovich_field(double scale_factor, const Config& config) {
    mesh_library_ready();
    const nbpm_ic_params p = ic_params(config, scale_factor);
    const std::size_t cells = static_cast<std::size_t>(config.MESH_SIZE) * config.MESH_SIZE * config.MESH_SIZE;
    // the one host-side array of the initial conditions: unit Gaussians in the order the standard
    // library's engine hands them out, one per mesh cell, z fastest
    std::vector<double> noise(cells);
    std::default_random_engine engine(config.SEED);
    std::normal_distribution<double> unit_gaussian(0.0, 1.0);
    for (double& v : noise) v = unit_gaussian(engine);
    DeviceArray white;
    white.upload(noise);
    ZeldovichField field;
    field.dx = DeviceArray(cells);
    field.dy = DeviceArray(cells);
    field.dz = DeviceArray(cells);
    nbpm_ok(nbpm_zeldovich_field_device(&p, white.data(), field.dx.data(), field.dy.data(), field.dz.data(), nullptr));
    hip_ok(hipDeviceSynchronize(), "sync");          // `white` is freed on return
    const double matter = config.OMEGA_M, lambda = config.OMEGA_LAMBDA * scale_factor * scale_factor * scale_factor;
    field.f = std::pow(matter / (matter + lambda), 0.55);
    return field;
}

DeviceParticles initialize_dm_device(const Cosmology& cosmo, const Config& config, const ZeldovichField& zf) {
    mesh_library_ready();
    nbpm_ic_params p = ic_params(config, cosmo.scale_factor);
    DeviceParticles out;
    out.count = static_cast<std::size_t>(config.N_PER_SIDE) * config.N_PER_SIDE * config.N_PER_SIDE;
    for (DeviceArray* a : {&out.x, &out.y, &out.z, &out.vx, &out.vy, &out.vz, &out.m}) *a = DeviceArray(out.count);
    const double velocity_per_displacement = config.STANDING_PARTICLES ? 0.0 : cosmo.hubble_param * zf.f;
    nbpm_ok(nbpm_zeldovich_particles_device(&p, zf.dx.data(), zf.dy.data(), zf.dz.data(), velocity_per_displacement,
                                            config.DM_PARTICLE_MASS, out.x.data(), out.y.data(), out.z.data(),
                                            out.vx.data(), out.vy.data(), out.vz.data(), out.m.data(), nullptr));
    hip_ok(hipDeviceSynchronize(), "sync");
    return out;
}

void initialize_dm(ParticleSystem& dm, const Cosmology& cosmo, const Config& config, const ZeldovichField& zf) {
    const DeviceParticles dev = initialize_dm_device(cosmo, config, zf);
    const std::vector<double> x = dev.x.download(), y = dev.y.download(), z = dev.z.download();
    const std::vector<double> vx = dev.vx.download(), vy = dev.vy.download(), vz = dev.vz.download();
    const std::vector<double> m = dev.m.download();
    for (std::size_t i = 0; i < dev.count; ++i) dm.add_particle(x[i], y[i], z[i], vx[i], vy[i], vz[i], m[i]);
}

DeviceParticles initialize_particles_device(const Config& config, Cosmology& cosmo, ZeldovichField* field_out) {
    cosmo.total_time = 0.0;
    update_cosmology(cosmo, config);
    ZeldovichField zf = compute_zeldovich_field(cosmo.scale_factor, config);
    DeviceParticles dev = initialize_dm_device(cosmo, config, zf);
    if (field_out) *field_out = std::move(zf);
    return dev;
}

ParticleSystem initialize_particles(const Config& config, Cosmology& cosmo, ZeldovichField* field_out) {
    cosmo.total_time = 0.0;
    update_cosmology(cosmo, config);
    ZeldovichField zf = compute_zeldovich_field(cosmo.scale_factor, config);
    ParticleSystem dm(config);
    initialize_dm(dm, cosmo, config, zf);
    if (field_out) *field_out = std::move(zf);
    return dm;
}

double internal_energy_from_temperature(double T_kelvin, double hubble_param, double box_size_mpc,
                                        double domain_size, double gamma) {
    // u = k_B T / ((gamma - 1) mu m_p) in m^2/s^2, over the square of the code's velocity unit
    // (1.5 H0 times the code's length unit: src/ics.cpp:218-239)
    constexpr double boltzmann = 1.380649e-23, proton_mass = 1.67262192e-27, mean_molecular_weight = 1.22;
    const double u_si = (boltzmann * T_kelvin) / ((gamma - 1.0) * mean_molecular_weight * proton_mass);
    const double H0_km_s_mpc = hubble_param * 100.0;
    const double length_unit_mpc = box_size_mpc / domain_size;
    const double velocity_unit_m_s = 1.5 * H0_km_s_mpc * length_unit_mpc * 1000.0;
    return u_si / (velocity_unit_m_s * velocity_unit_m_s);
}

}  // namespace nbh
