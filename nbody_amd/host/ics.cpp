#include "ics.hpp"

#include <utility>

#include <hip/hip_runtime_api.h>
#include <hipfft/hipfft.h>

#include <cmath>
#include <complex>
#include <random>
#include <stdexcept>
#include <string>

namespace nbh {
namespace {

// BBKS transfer-function shape, unnormalised (src/ics.cpp:11-27).
double unnormalized_pk(double k_h, const Config& config) {
    const double Gamma = config.OMEGA_M * config.HUBBLE_PARAM;
    const double q = k_h / Gamma;
    const double T_k = std::log(1.0 + 2.34 * q) / (2.34 * q) *
                       std::pow(1.0 + 3.89 * q + std::pow(16.1 * q, 2) + std::pow(5.46 * q, 3) +
                                    std::pow(6.71 * q, 4),
                                -0.25);
    return std::pow(k_h, config.SPECTRAL_INDEX) * T_k * T_k;
}

double window_tophat(double kR) {  // src/ics.cpp:30-34
    if (kR < 1e-4) return 1.0 - (kR * kR / 10.0);
    return 3.0 * (std::sin(kR) - kR * std::cos(kR)) / (kR * kR * kR);
}

void hip_ok(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void fft_ok(hipfftResult r, const char* what) {
    if (r != HIPFFT_SUCCESS) throw std::runtime_error(std::string(what) + ": hipFFT error " + std::to_string(r));
}

// Unnormalised 3-D transforms of an M^3 real field, z fastest (the layout of src/ics.cpp:68-79).
class Fft3d {
   public:
    explicit Fft3d(int M) : M_(M), n_real_(static_cast<size_t>(M) * M * M),
                            n_cplx_(static_cast<size_t>(M) * M * (M / 2 + 1)) {
        hip_ok(hipMalloc(&real_, n_real_ * sizeof(double)), "hipMalloc");
        hip_ok(hipMalloc(&cplx_, n_cplx_ * sizeof(hipfftDoubleComplex)), "hipMalloc");
        fft_ok(hipfftPlan3d(&fwd_, M, M, M, HIPFFT_D2Z), "hipfftPlan3d");
        fft_ok(hipfftPlan3d(&inv_, M, M, M, HIPFFT_Z2D), "hipfftPlan3d");
    }
    ~Fft3d() {
        hipfftDestroy(fwd_);
        hipfftDestroy(inv_);
        (void)hipFree(real_);
        (void)hipFree(cplx_);
    }
    void forward(const std::vector<double>& in, std::vector<std::complex<double>>& out) {
        hip_ok(hipMemcpy(real_, in.data(), n_real_ * sizeof(double), hipMemcpyHostToDevice), "upload");
        fft_ok(hipfftExecD2Z(fwd_, real_, cplx_), "hipfftExecD2Z");
        out.resize(n_cplx_);
        hip_ok(hipMemcpy(out.data(), cplx_, n_cplx_ * sizeof(hipfftDoubleComplex), hipMemcpyDeviceToHost), "download");
    }
    void inverse(const std::vector<std::complex<double>>& in, std::vector<double>& out) {
        hip_ok(hipMemcpy(cplx_, in.data(), n_cplx_ * sizeof(hipfftDoubleComplex), hipMemcpyHostToDevice), "upload");
        fft_ok(hipfftExecZ2D(inv_, cplx_, real_), "hipfftExecZ2D");
        out.resize(n_real_);
        hip_ok(hipMemcpy(out.data(), real_, n_real_ * sizeof(double), hipMemcpyDeviceToHost), "download");
    }

   private:
    int M_;
    size_t n_real_, n_cplx_;
    double* real_ = nullptr;
    hipfftDoubleComplex* cplx_ = nullptr;
    hipfftHandle fwd_ = 0, inv_ = 0;
};

}  // namespace

ZeldovichField compute_zeldovich_field(double scale_factor, const Config& config) {
    // sigma_8 normalisation: the same quadrature, the same running k (src/ics.cpp:38-51)
    const double R = 8.0, dk = 0.001;
    double unnorm_variance = 0.0;
    for (double k = 0.001; k < 100.0; k += dk) {
        const double pk = unnormalized_pk(k, config);
        const double w = window_tophat(k * R);
        unnorm_variance += pk * w * w * k * k * dk;
    }
    unnorm_variance /= (2.0 * M_PI * M_PI);
    const double A = (config.SIGMA_8 * config.SIGMA_8) / unnorm_variance;

    const int M = config.MESH_SIZE;
    const size_t M3 = static_cast<size_t>(M) * M * M;
    const double box_vol = std::pow(config.BOX_SIZE_MPC, 3.0);
    const double amplitude_scaling = std::sqrt(static_cast<double>(M3) / box_vol);   // :57-59

    std::vector<double> white(M3);                                                   // :61-66
    std::default_random_engine generator(config.SEED);
    std::normal_distribution<double> distribution(0.0, 1.0);
    for (auto& v : white) v = distribution(generator);

    Fft3d fft(M);
    std::vector<std::complex<double>> random_k;
    fft.forward(white, random_k);                                                    // :77-79

    const size_t half = static_cast<size_t>(M / 2 + 1);
    std::vector<std::complex<double>> kx_k(random_k.size()), ky_k(random_k.size()), kz_k(random_k.size());
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j)
            for (int k = 0; k < M / 2 + 1; ++k) {
                const size_t idx = (static_cast<size_t>(i) * M + j) * half + k;
                if (i == 0 && j == 0 && k == 0) {                                    // :88-93
                    kx_k[0] = ky_k[0] = kz_k[0] = {0.0, 0.0};
                    continue;
                }
                const double fx = (i < M / 2) ? i : (i - M);                         // :95-97
                const double fy = (j < M / 2) ? j : (j - M);
                const double fz = k;
                const double kx = fx * 2.0 * M_PI / config.BOX_SIZE_MPC;             // :103-105
                const double ky = fy * 2.0 * M_PI / config.BOX_SIZE_MPC;
                const double kz = fz * 2.0 * M_PI / config.BOX_SIZE_MPC;
                const double k_mag = std::sqrt(kx * kx + ky * ky + kz * kz);
                const double k_h = k_mag / config.HUBBLE_PARAM;
                double power_spectrum_sqrt = 0.0;
                if (k_mag > 0.0)
                    power_spectrum_sqrt = std::sqrt(A * unnormalized_pk(k_h, config)) * amplitude_scaling;
                const std::complex<double> delta_k = random_k[idx] * power_spectrum_sqrt;   // :118-119
                const double cx = fx * 2.0 * M_PI / config.DOMAIN_SIZE;              // :123-125
                const double cy = fy * 2.0 * M_PI / config.DOMAIN_SIZE;
                const double cz = fz * 2.0 * M_PI / config.DOMAIN_SIZE;
                const double code_k2 = cx * cx + cy * cy + cz * cz;
                const std::complex<double> phi_k = -delta_k / code_k2;               // :129
                kx_k[idx] = std::complex<double>(0, -1) * cx * phi_k;                // :131-133
                ky_k[idx] = std::complex<double>(0, -1) * cy * phi_k;
                kz_k[idx] = std::complex<double>(0, -1) * cz * phi_k;
            }

    ZeldovichField field;
    fft.inverse(kx_k, field.dx);                                                     // :143-148
    fft.inverse(ky_k, field.dy);
    fft.inverse(kz_k, field.dz);
    const double norm_ic = scale_factor / static_cast<double>(M3);                   // :150-156
    for (size_t i = 0; i < M3; ++i) {
        field.dx[i] *= norm_ic;
        field.dy[i] *= norm_ic;
        field.dz[i] *= norm_ic;
    }
    const double a3 = std::pow(scale_factor, 3.0);                                   // :159-162
    const double Om_a = config.OMEGA_M / (config.OMEGA_M + config.OMEGA_LAMBDA * a3);
    field.f = std::pow(Om_a, 0.55);
    return field;
}

void initialize_dm(ParticleSystem& dm, const Cosmology& cosmo, const Config& config,
                   const ZeldovichField& zf) {
    const int M = config.MESH_SIZE;
    const double cell_size = config.DOMAIN_SIZE / M;
    const int N_part = config.N_PER_SIDE;
    const double spacing = config.DOMAIN_SIZE / N_part;
    for (int i = 0; i < N_part; ++i)
        for (int j = 0; j < N_part; ++j)
            for (int k = 0; k < N_part; ++k) {
                const double qx = (i + 0.5) * spacing, qy = (j + 0.5) * spacing, qz = (k + 0.5) * spacing;
                const int ix = static_cast<int>(qx / cell_size) % M;                 // :181-183
                const int iy = static_cast<int>(qy / cell_size) % M;
                const int iz = static_cast<int>(qz / cell_size) % M;
                const size_t idx = (static_cast<size_t>(ix) * M + iy) * M + iz;      // :185-187
                const double dx = zf.dx[idx], dy = zf.dy[idx], dz = zf.dz[idx];
                const double px = std::fmod(qx + dx + config.DOMAIN_SIZE, config.DOMAIN_SIZE);   // :193-198
                const double py = std::fmod(qy + dy + config.DOMAIN_SIZE, config.DOMAIN_SIZE);
                const double pz = std::fmod(qz + dz + config.DOMAIN_SIZE, config.DOMAIN_SIZE);
                const bool still = config.STANDING_PARTICLES;                        // :200-208
                dm.add_particle(px, py, pz, still ? 0.0 : cosmo.hubble_param * dx * zf.f,
                                still ? 0.0 : cosmo.hubble_param * dy * zf.f,
                                still ? 0.0 : cosmo.hubble_param * dz * zf.f, config.DM_PARTICLE_MASS);
            }
}

ParticleSystem initialize_particles(const Config& config, Cosmology& cosmo, ZeldovichField* field_out) {
    cosmo.total_time = 0.0;                                                          // :292-293
    update_cosmology(cosmo, config);
    ZeldovichField zf = compute_zeldovich_field(cosmo.scale_factor, config);         // :295-296
    ParticleSystem dm(config);
    initialize_dm(dm, cosmo, config, zf);                                            // :299
    if (field_out) *field_out = std::move(zf);
    return dm;
}

double internal_energy_from_temperature(double T_kelvin, double hubble_param, double box_size_mpc,
                                        double domain_size, double gamma) {
    const double k_B = 1.380649e-23, m_p = 1.67262192e-27, mu = 1.22;
    const double u_physical_m2_s2 = (k_B * T_kelvin) / ((gamma - 1.0) * mu * m_p);
    const double H0 = hubble_param * 100.0;
    const double code_unit_mpc = box_size_mpc / domain_size;
    const double V_unit_km_s = 1.5 * H0 * code_unit_mpc;
    const double V_unit_m_s = V_unit_km_s * 1000.0;
    return u_physical_m2_s2 / (V_unit_m_s * V_unit_m_s);
}

}  // namespace nbh
