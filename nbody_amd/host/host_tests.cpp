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
//   nbody_host_tests run <ini> <cycles>    ICs + device-resident engine for <cycles> steps; started
//                                          once per GPU by a launcher that sets WORLD_SIZE, RANK
//                                          and LOCAL_RANK, the processes share the particles
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
#include "simulation.hpp"
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
            // run <ini> <cycles> [snapshot directory (HDF5) or file prefix (raw) [diagnostics.csv
            // directory [cycles between device-time samples]]]
            // one process per GPU when a launcher set WORLD_SIZE / RANK / LOCAL_RANK (libnbmg),
            // else GPU 0 alone
            if (std::getenv("WORLD_SIZE")) {
                if (nbmg_init_from_env() != NBMG_OK) { std::printf("nbmg_init_from_env failed: %s\n", nbmg_last_error()); return 2; }
            } else if (nbpp_init(0) != NBPP_OK) { std::printf("nbpp_init failed: %s\n", nbpp_last_error()); return 2; }
            Config config;
            config.load(argv[2]);
            config.MAX_CYCLES = std::atoi(argv[3]);
            RunOptions options;
            if (argc >= 5) options.snapshot_target = argv[4];
            if (argc >= 6) options.log_dir = argv[5];
            if (argc >= 7) options.timing_every = std::atoi(argv[6]);
            if (std::getenv("NBODY_NO_OVERLAP")) options.overlap_forces = false;   // A/B switch
            Simulation sim(config, options);
            sim.report("initial");
            std::signal(SIGINT, on_sigint);
            const ExitStatus st = sim.run(&g_stop_requested);
            sim.report("final");
            sim.print_timing(st);
            nbgas_shutdown();
            nbpm_shutdown();
            nbmg_shutdown();
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
