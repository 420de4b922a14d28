// host_tests.cpp - driver and self-tests for the C++ host side (Config / ParticleSystem / KDK)
// over libnbpp.  The GPU cases restate the reference's own unit tests for this path
// (tests/test_gravity.cpp:98-155) against the mirrored interface.
//
//   nbody_host_tests config <file.ini>     print every Config field (hex floats), no GPU needed
//   nbody_host_tests gpu-tests             known-answer tests on GPU 0, exit code = failures
//   nbody_host_tests kdk <in.bin> <out.bin> <steps> <dt> <ini>
//                                          PP-only KDK steps on particles read from in.bin
//                                          (dt <= 0: the engine's adaptive time step)
//   nbody_host_tests ics <ini> <out.bin>   Zel'dovich particles of <ini> (x y z vx vy vz), GPU FFT
//   nbody_host_tests run <ini> <cycles>    ICs + device-resident engine for <cycles> steps
//   nbody_host_tests engine <in.bin> <out.bin> <steps> <dt> <ini>
//                                          the same steps with the state resident on the GPU
#include <chrono>
#include <csignal>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "config.hpp"
#include "device_engine.hpp"
#include "diagnostics.hpp"
#include "ics.hpp"
#include "integrator.hpp"
#include "particles.hpp"
#include "snapshot.hpp"

using namespace nbh;

static int g_failures = 0;
#define REQUIRE(cond)                                                         \
    do {                                                                      \
        if (!(cond)) {                                                        \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);       \
            ++g_failures;                                                     \
        }                                                                     \
    } while (0)

static bool approx(double got, double want, double rel = 1e-12, double margin = 0.0) {
    return std::fabs(got - want) <= std::max(margin, rel * std::fabs(want));
}

static void print_config(const Config& c) {
#define D(name) std::printf(#name "=%a\n", static_cast<double>(c.name))
#define I(name) std::printf(#name "=%d\n", static_cast<int>(c.name))
    D(DOMAIN_SIZE); I(MESH_SIZE); D(BOX_SIZE_MPC); D(OMEGA_BARYON); D(OMEGA_M); D(OMEGA_LAMBDA);
    D(HUBBLE_PARAM); I(EXPANDING_UNIVERSE); I(N_PER_SIDE); I(STANDING_PARTICLES); D(START_A);
    D(SIGMA_8); D(SPECTRAL_INDEX); D(INITIAL_GAS_TEMPERATURE_K); I(SEED); I(USE_HYDRO); D(GAMMA);
    I(USE_PM); I(USE_PP); D(CUTOFF_RADIUS_CELLS); D(CUTOFF_TRANSITION_WIDTH_FACTOR); D(DT_FACTOR);
    D(MAX_SCALE_FACTOR); D(CFL_SAFETY_FACTOR); D(GRAVITY_DT_FACTOR); I(USE_ADAPTIVE_DT);
    I(MAX_CYCLES); D(SAVE_HDF5_EVERY_DELTA_A); I(DEBUG_INFO_EVERY_CYCLES);
    I(ENABLE_ENERGY_DIAGNOSTICS); D(G); D(CELL_SIZE); D(CELL_VOLUME); D(OMEGA_DM);
    D(CUTOFF_RADIUS); D(CUTOFF_RADIUS_SQUARED); D(CUTOFF_TRANSITION_WIDTH); D(R_SWITCH_START);
    D(R_SWITCH_START_SQ); I(NUM_DM_PARTICLES); D(DM_PARTICLE_MASS); D(GAS_TOTAL_MASS);
    D(SOFTENING_SQUARED); D(FIXED_DT);
#undef D
#undef I
}

// tests/test_gravity.cpp:98-113: the shared two-body set-up.
static Config two_body_config() {
    Config config;
    config.MESH_SIZE = 4;
    config.CELL_SIZE = 2.0;
    config.DOMAIN_SIZE = 8.0;
    config.NUM_DM_PARTICLES = 2;
    config.G = 1.0;
    config.SOFTENING_SQUARED = 0.0;
    return config;
}

static void test_pure_newtonian() {  // tests/test_gravity.cpp:115-134
    Config config = two_body_config();
    ParticleSystem sys(config);
    sys.add_particle(2.0, 2.0, 2.0, 0.0, 0.0, 0.0, 1.0);
    sys.add_particle(5.0, 2.0, 2.0, 0.0, 0.0, 0.0, 1.0);
    config.USE_PM = false;
    config.CUTOFF_RADIUS_SQUARED = (config.DOMAIN_SIZE / 2.0) * (config.DOMAIN_SIZE / 2.0);
    sys.bin_and_assign_mass(config);
    sys.compute_pp_forces(config);
    const double expected_f = 1.0 / 9.0;
    REQUIRE(approx(sys.acc_x[0], expected_f));
    REQUIRE(sys.acc_y[0] == 0.0);
    REQUIRE(sys.acc_z[0] == 0.0);
    REQUIRE(approx(sys.acc_x[1], -expected_f));
}

static void test_p3m_cutoff() {  // tests/test_gravity.cpp:136-154
    Config config = two_body_config();
    ParticleSystem sys(config);
    sys.add_particle(2.0, 2.0, 2.0, 0.0, 0.0, 0.0, 1.0);
    sys.add_particle(5.0, 2.0, 2.0, 0.0, 0.0, 0.0, 1.0);
    config.USE_PM = true;
    config.CUTOFF_RADIUS = 2.0;
    config.CUTOFF_RADIUS_SQUARED = 4.0;
    config.R_SWITCH_START = 1.0;
    config.R_SWITCH_START_SQ = 1.0;
    config.CUTOFF_TRANSITION_WIDTH = 1.0;
    sys.bin_and_assign_mass(config);
    std::fill(sys.acc_x.begin(), sys.acc_x.end(), 0.0);
    sys.compute_pp_forces(config);
    REQUIRE(sys.acc_x[0] == 0.0);
    REQUIRE(sys.acc_x[1] == 0.0);
}

static void test_wrap_through_forces() {  // tests/test_math.cpp:14-37 seen through the force
    Config config;
    config.DOMAIN_SIZE = 64.0;
    config.MESH_SIZE = 8;
    config.N_PER_SIDE = 2;
    config.USE_PM = false;
    config.compute_derived_data();
    config.G = 1.0;
    config.SOFTENING_SQUARED = 0.0;
    ParticleSystem sys(config);
    sys.add_particle(1.0, 5.0, 5.0, 0, 0, 0, 1.0);   // A at x = 1
    sys.add_particle(63.0, 5.0, 5.0, 0, 0, 0, 1.0);  // B at x = 63: shortest path is -2
    sys.compute_pp_forces(config);
    REQUIRE(approx(sys.acc_x[0], -0.25));
    REQUIRE(approx(sys.acc_x[1], +0.25));
}

static void test_accumulate_and_contract() {
    Config config;
    config.MESH_SIZE = 8;
    config.N_PER_SIDE = 6;
    config.USE_PM = false;
    config.compute_derived_data();
    ParticleSystem sys(config);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) / 16777216.0; };
    for (int i = 0; i < config.NUM_DM_PARTICLES; ++i)
        sys.add_particle(rnd(), rnd(), rnd(), 0, 0, 0, config.DM_PARTICLE_MASS);
    sys.compute_pp_forces(config);
    std::vector<double> once = sys.acc_x;
    sys.compute_pp_forces(config);  // acc += : second call doubles
    for (std::size_t i = 0; i < sys.num_particles; ++i) REQUIRE(approx(sys.acc_x[i], 2.0 * once[i], 1e-14));
    compute_accelerations(sys, config);  // zero-fills first
    double mx = 1e-9;
    for (std::size_t i = 0; i < sys.num_particles; ++i) {
        REQUIRE(approx(sys.acc_x[i], once[i], 1e-14));
        mx = std::max(mx, sys.acc_x[i] * sys.acc_x[i] + sys.acc_y[i] * sys.acc_y[i] + sys.acc_z[i] * sys.acc_z[i]);
    }
    REQUIRE(sys.max_accel_sq == mx);
    REQUIRE(sys.get_gravity_timestep(config) ==
            std::sqrt(config.SOFTENING_SQUARED) / std::sqrt(mx) * config.GRAVITY_DT_FACTOR);
    config.STANDING_PARTICLES = true;  // src/integrator.cpp:263-269
    compute_accelerations(sys, config);
    REQUIRE(sys.max_accel_sq == 1e-9);
    for (std::size_t i = 0; i < sys.num_particles; ++i) REQUIRE(sys.acc_x[i] == 0.0);
    config.STANDING_PARTICLES = false;
    config.USE_PP = false;             // src/integrator.cpp:254
    compute_accelerations(sys, config);
    REQUIRE(sys.max_accel_sq == 1e-9);
    // total momentum change of a KDK step is zero (pairwise antisymmetry)
    config.USE_PP = true;
    config.EXPANDING_UNIVERSE = false;
    Cosmology cosmo;
    compute_accelerations(sys, config);
    KDK_step(sys, cosmo, config.FIXED_DT, config);
    double px = 0, scale = 0;
    for (std::size_t i = 0; i < sys.num_particles; ++i) {
        px += sys.mass[i] * sys.vel_x[i];
        scale += sys.mass[i] * std::fabs(sys.vel_x[i]);
        REQUIRE(sys.pos_x[i] >= 0.0 && sys.pos_x[i] <= config.DOMAIN_SIZE);
    }
    REQUIRE(std::fabs(px) < 1e-10 * scale);
    ParticleSystem empty(config);
    empty.compute_pp_forces(config);  // n == 0: no-op
    REQUIRE(std::isinf(empty.get_gravity_timestep(config)));
}

static int run_kdk(const char* in_path, const char* out_path, int steps, double dt, const char* ini,
                   bool device_resident) {
    Config config;
    config.load(ini);
    config.USE_HYDRO = false;   // the particle file carries no gas: dark matter only
    config.MAX_CYCLES = steps;
    config.USE_ADAPTIVE_DT = !(dt > 0.0);
    if (dt > 0.0) config.FIXED_DT = dt;
    FILE* f = std::fopen(in_path, "rb");
    if (!f) throw std::runtime_error("cannot open input");
    long long n = 0;
    if (std::fread(&n, sizeof(n), 1, f) != 1) throw std::runtime_error("short read");
    std::vector<double> buf(7 * n);
    if (std::fread(buf.data(), sizeof(double), buf.size(), f) != buf.size()) throw std::runtime_error("short read");
    std::fclose(f);
    ParticleSystem sys(config);
    for (long long i = 0; i < n; ++i)
        sys.add_particle(buf[i], buf[n + i], buf[2 * n + i], buf[3 * n + i], buf[4 * n + i],
                         buf[5 * n + i], buf[6 * n + i]);
    Cosmology cosmo;
    double last_dt = 0.0;
    if (device_resident) {
        nbpp_set_option(NBPP_OPT_TIMING, 1);
        DeviceEngine engine(config, sys);
        while (engine.cycle_count < steps) { last_dt = engine.current_dt; engine.step(); }
        engine.download(sys);
        cosmo = engine.cosmo;
        std::printf("device-resident: %ld cycles, PP %.3f ms total\n", engine.cycle_count, engine.pp_ms);
    } else {
        update_cosmology(cosmo, config);
        compute_accelerations(sys, config);  // initial accelerations, as src/ics.cpp:305-346
        double cur = config.USE_ADAPTIVE_DT ? std::min(sys.get_gravity_timestep(config), config.FIXED_DT)
                                            : config.FIXED_DT;
        for (int s = 0; s < steps; ++s) {
            last_dt = cur;
            KDK_step(sys, cosmo, cur, config);
            if (config.USE_ADAPTIVE_DT) cur = std::min(sys.get_gravity_timestep(config), config.FIXED_DT);
        }
    }
    f = std::fopen(out_path, "wb");
    if (!f) throw std::runtime_error("cannot open output");
    std::fwrite(&n, sizeof(n), 1, f);
    for (auto* v : {&sys.pos_x, &sys.pos_y, &sys.pos_z, &sys.vel_x, &sys.vel_y, &sys.vel_z,
                    &sys.acc_x, &sys.acc_y, &sys.acc_z})
        std::fwrite(v->data(), sizeof(double), n, f);
    const double tail[5] = {sys.max_accel_sq, cosmo.scale_factor, cosmo.hubble_param, cosmo.total_time, last_dt};
    std::fwrite(tail, sizeof(double), 5, f);
    std::fclose(f);
    return 0;
}

// Ctrl+C finishes the cycle in flight, writes the last snapshot and diagnostics line and ends the
// run with exit=2 (UserAborted), as src/main.cpp:13-22 and src/engine.cpp:79 do.
volatile std::sig_atomic_t g_stop_requested = 0;
extern "C" void on_sigint(int) { g_stop_requested = 1; }

int main(int argc, char** argv) {
    try {
        if (argc >= 2 && !std::strcmp(argv[1], "config")) {
            Config c;
            if (argc >= 3) c.load(argv[2]);
            print_config(c);
            return 0;
        }
        if (argc >= 4 && !std::strcmp(argv[1], "ics")) {
            Config config;
            config.load(argv[2]);
            Cosmology cosmo;
            ParticleSystem dm = initialize_particles(config, cosmo);
            FILE* f = std::fopen(argv[3], "wb");
            if (!f) throw std::runtime_error("cannot open output");
            long long n = static_cast<long long>(dm.num_particles);
            std::fwrite(&n, sizeof(n), 1, f);
            for (auto* v : {&dm.pos_x, &dm.pos_y, &dm.pos_z, &dm.vel_x, &dm.vel_y, &dm.vel_z})
                std::fwrite(v->data(), sizeof(double), n, f);
            const double tail[2] = {cosmo.scale_factor, cosmo.hubble_param};
            std::fwrite(tail, sizeof(double), 2, f);
            std::fclose(f);
            return 0;
        }
        if (argc >= 4 && !std::strcmp(argv[1], "snapshot")) {
            // a patterned state of the size <ini> asks for as snapshot_0007.hdf5 in <dir>: no GPU
            // involved.  Particle i sits at ((i + 0.25) / n, (i + 0.5) / n, (i + 0.75) / n) L with
            // velocity (i, -i, 2 i) and acceleration (0.5 i, 0, -0.5 i); gas mesh k holds
            // k + 1 + cell index / 1024.
            Config config;
            config.load(argv[2]);
            ParticleSystem dm(config);
            const int n = config.NUM_DM_PARTICLES;
            for (int i = 0; i < n; ++i) {
                dm.add_particle((i + 0.25) / n * config.DOMAIN_SIZE, (i + 0.5) / n * config.DOMAIN_SIZE,
                                (i + 0.75) / n * config.DOMAIN_SIZE, i, -i, 2.0 * i, config.DM_PARTICLE_MASS);
                dm.acc_x[i] = 0.5 * i;
                dm.acc_z[i] = -0.5 * i;
            }
            std::vector<double> gas[9];
            if (config.USE_HYDRO) {
                const std::size_t cells = static_cast<std::size_t>(config.MESH_SIZE) * config.MESH_SIZE * config.MESH_SIZE;
                for (int k = 0; k < 9; ++k) {
                    gas[k].resize(cells);
                    for (std::size_t c = 0; c < cells; ++c) gas[k][c] = k + 1 + static_cast<double>(c) / 1024.0;
                }
            }
            SnapshotWriter writer(argv[3], config);
            writer.save_snapshot(7, 0, dm, gas, 1.25, config.START_A, config);
            std::printf("wrote %s\n", writer.path_of(7).c_str());
            return 0;
        }
        if (argc >= 4 && !std::strcmp(argv[1], "run")) {
            if (nbpp_init(0) != NBPP_OK) { std::printf("nbpp_init failed: %s\n", nbpp_last_error()); return 2; }
            nbpp_set_option(NBPP_OPT_TIMING, 1);
            Config config;
            config.load(argv[2]);
            config.MAX_CYCLES = std::atoi(argv[3]);
            Cosmology cosmo;
            ZeldovichField field;
            ParticleSystem dm = initialize_particles(config, cosmo, &field);
            DeviceEngine engine(config, dm, config.USE_HYDRO ? &field : nullptr);
            // device times are sampled on the cycles that print them, and on the last one of a run
            // bounded by its cycle count (optional sixth argument: another sampling interval, 1 =
            // every cycle)
            engine.timing_every = argc >= 7 ? std::atoi(argv[6])
                                            : std::max(1, std::min(config.DEBUG_INFO_EVERY_CYCLES, config.MAX_CYCLES));
            auto report = [&](const char* when) {
                ParticleSystem now(config);
                engine.download(now);
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
                    std::printf("%s: gas mass %.17g energy %.17g\n", when, mass * config.CELL_VOLUME,
                                energy * config.CELL_VOLUME);
                }
            };
            report("initial");
            // the first force phase also pays for set-up (FFT plans, buffers): keep it out of the averages
            engine.pp_ms = engine.pm_ms = engine.gas_ms = 0.0;
            engine.timed_force_calls = engine.timed_steps = 0;
            // Snapshots with the reference's cadence (src/engine.cpp:28-33, :56-67): one at the start,
            // one whenever a passes next_output_a, one at the end.  If the fourth argument names a
            // directory: snapshot_%04d.hdf5 files in it, the reference's layout (snapshot.hpp).
            // Otherwise it is a file prefix for raw little-endian doubles:
            // {n, mesh, a, t, L, use_hydro}, x, y, z, mass[n], then gas density and energy [mesh^3]
            // with hydro.
            const char* snap_prefix = argc >= 5 ? argv[4] : nullptr;
            std::unique_ptr<SnapshotWriter> hdf5;
            if (snap_prefix && std::filesystem::is_directory(snap_prefix))
                hdf5.reset(new SnapshotWriter(snap_prefix, config));
            int snapshot_count = 0;
            double next_output_a = config.START_A + config.SAVE_HDF5_EVERY_DELTA_A;
            auto save_snapshot = [&]() {
                if (!snap_prefix) return;
                ParticleSystem now(config);
                engine.download(now);
                std::vector<double> gas[9];
                engine.download_gas(gas);
                if (hdf5) {
                    hdf5->save_snapshot(snapshot_count++, engine.cycle_count, now, gas, engine.cosmo.total_time,
                                        engine.cosmo.scale_factor, config);
                    return;
                }
                char name[1024];
                std::snprintf(name, sizeof(name), "%s_%04d.bin", snap_prefix, snapshot_count++);
                FILE* f = std::fopen(name, "wb");
                if (!f) throw std::runtime_error("cannot open snapshot file");
                const double head[6] = {static_cast<double>(now.num_particles), static_cast<double>(config.MESH_SIZE),
                                        engine.cosmo.scale_factor, engine.cosmo.total_time, config.DOMAIN_SIZE,
                                        config.USE_HYDRO ? 1.0 : 0.0};
                std::fwrite(head, sizeof(double), 6, f);
                for (auto* v : {&now.pos_x, &now.pos_y, &now.pos_z, &now.mass})
                    std::fwrite(v->data(), sizeof(double), v->size(), f);
                if (config.USE_HYDRO) {
                    std::fwrite(gas[0].data(), sizeof(double), gas[0].size(), f);
                    std::fwrite(gas[4].data(), sizeof(double), gas[4].size(), f);
                }
                std::fclose(f);
            };
            if (config.SAVE_HDF5_EVERY_DELTA_A > 0.0) save_snapshot();
            // diagnostics.csv in the directory given as the fifth argument, with the reference's
            // columns and cadence (src/engine.cpp:24-26, :69-75)
            Logger logger(argc >= 6 ? argv[5] : "/nonexistent");
            long logged_cycle = 0, logged_calls = 0, logged_steps = 0;
            double logged_ms[3] = {0.0, 0.0, 0.0}, last_avg[3] = {0.0, 0.0, 0.0};
            auto logged_wall = std::chrono::steady_clock::now();
            auto log_diagnostics = [&]() {
                if (!logger.is_open()) return;
                ParticleSystem now(config);
                engine.download(now);
                std::vector<double> gas[9], total_rho, phi;
                engine.download_gas(gas);
                engine.download_mesh_solve(total_rho, phi);
                Diagnostics d = compute_diagnostics(config, now, gas, total_rho, phi, engine.cosmo,
                                                    engine.cfl_timestep(), engine.gravity_timestep(), engine.current_dt);
                d.cycle = engine.cycle_count;
                const long cycles = engine.cycle_count - logged_cycle;
                const auto wall_now = std::chrono::steady_clock::now();
                if (cycles > 0) {                                    // averages since the last line
                    d.avg_step = std::chrono::duration<double>(wall_now - logged_wall).count() / cycles;
                    // device times: averages over the cycles since the last line that were timed
                    // (none timed, as can happen for the last line of a run: the averages before it)
                    const long calls = engine.timed_force_calls - logged_calls;
                    const long steps = engine.timed_steps - logged_steps;
                    if (calls > 0) {
                        last_avg[0] = (engine.pm_ms - logged_ms[0]) * 1e-3 / calls;
                        last_avg[1] = (engine.pp_ms - logged_ms[1]) * 1e-3 / calls;
                    }
                    if (steps > 0) last_avg[2] = (engine.gas_ms - logged_ms[2]) * 1e-3 / steps;
                    d.avg_pm = last_avg[0];
                    d.avg_pp = last_avg[1];
                    d.avg_hydro = last_avg[2];
                }
                logger.log(d);
                logged_cycle = engine.cycle_count;
                logged_ms[0] = engine.pm_ms; logged_ms[1] = engine.pp_ms; logged_ms[2] = engine.gas_ms;
                logged_calls = engine.timed_force_calls; logged_steps = engine.timed_steps;
                logged_wall = std::chrono::steady_clock::now();
            };
            log_diagnostics();
            const auto t0 = std::chrono::steady_clock::now();
            // engine.run() (src/engine.cpp:78-91) unrolled, to log like the reference does every
            // DEBUG_INFO_EVERY_CYCLES cycles (src/engine.cpp:69-75)
            std::signal(SIGINT, on_sigint);
            while (engine.needs_more_cycles && !g_stop_requested) {
                engine.step();
                if (config.DEBUG_INFO_EVERY_CYCLES > 0 && engine.cycle_count % config.DEBUG_INFO_EVERY_CYCLES == 0) {
                    nbpp_stats ps{};
                    nbpp_get_stats(&ps);
                    std::printf("cycle %ld: a=%.6f dt=%.4e max|a|^2=%.4e | last PP call %.3f ms (binning %.3f, pairs %.3f), %.4g accepted pairs (%.1f G/s)\n",
                                engine.cycle_count, engine.cosmo.scale_factor, engine.current_dt, engine.max_accel_sq,
                                ps.total_ms, ps.prepare_ms, ps.pair_kernel_ms, static_cast<double>(ps.pairs_accepted),
                                ps.total_ms > 0 ? ps.pairs_accepted / ps.total_ms * 1e-6 : 0.0);
                    std::fflush(stdout);
                }
                const bool last_cycle = !engine.needs_more_cycles || g_stop_requested;
                if (config.DEBUG_INFO_EVERY_CYCLES > 0 &&
                    (last_cycle || engine.cycle_count % config.DEBUG_INFO_EVERY_CYCLES == 0))
                    log_diagnostics();
                const bool must_save = engine.cosmo.scale_factor >= next_output_a;
                if (config.SAVE_HDF5_EVERY_DELTA_A > 0.0 && (last_cycle || must_save)) {
                    save_snapshot();
                    while (next_output_a <= engine.cosmo.scale_factor) next_output_a += config.SAVE_HDF5_EVERY_DELTA_A;
                }
            }
            const ExitStatus st = g_stop_requested ? ExitStatus::UserAborted
                                  : engine.cycle_count >= config.MAX_CYCLES ? ExitStatus::ReachedMaxCycles
                                                                            : ExitStatus::ReachedMaxScaleFactor;
            const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            report("final");
            std::printf("timing: %ld cycles, %.3f ms/cycle wall; PP %.3f, mesh %.3f, hydro sweeps %.3f ms/cycle (device, %ld timed cycles); exit=%d\n",
                        engine.cycle_count, wall_ms / std::max(1L, engine.cycle_count),
                        engine.pp_ms / std::max(1L, engine.timed_force_calls), engine.pm_ms / std::max(1L, engine.timed_force_calls),
                        engine.gas_ms / std::max(1L, engine.timed_steps), engine.timed_steps, static_cast<int>(st));
            nbgas_shutdown();
            nbpm_shutdown();
            nbpp_shutdown();
            return 0;
        }
        if (argc >= 2 && (!std::strcmp(argv[1], "gpu-tests") || !std::strcmp(argv[1], "kdk") ||
                          !std::strcmp(argv[1], "engine"))) {
            if (nbpp_init(0) != NBPP_OK) {
                std::printf("nbpp_init failed: %s\n", nbpp_last_error());
                return 2;
            }
            int rc = 0;
            if (std::strcmp(argv[1], "gpu-tests")) {
                if (argc < 7) throw std::runtime_error("needs <in> <out> <steps> <dt> <ini>");
                rc = run_kdk(argv[2], argv[3], std::atoi(argv[4]), std::atof(argv[5]), argv[6],
                             !std::strcmp(argv[1], "engine"));
            } else {
                test_pure_newtonian();
                test_p3m_cutoff();
                test_wrap_through_forces();
                test_accumulate_and_contract();
                std::printf("%s (%d failures)\n", g_failures ? "FAILED" : "ALL PASSED", g_failures);
                rc = g_failures ? 1 : 0;
            }
            nbpm_shutdown();
            nbpp_shutdown();
            return rc;
        }
        std::printf("usage: %s config [ini] | snapshot <ini> <dir> | gpu-tests | kdk <in> <out> <steps> <dt> <ini> | run <ini> <cycles> [snapshot-prefix|snapshot-dir [log-dir [timing-every]]]\n", argv[0]);
        return 64;
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 3;
    }
}
