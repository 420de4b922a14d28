// main.cpp - the `nbody` executable: the reference's command line (src/main.cpp:24-90) over the
// device-resident engine.
//
//   nbody [simulation.ini]
//
// Loads the configuration (default ./simulation.ini), creates outputs/run_<timestamp>/, copies
// the configuration there, and runs to MAX_SCALE_FACTOR or MAX_CYCLES writing snapshot_%04d.hdf5
// and diagnostics.csv into that directory.  Ctrl+C finishes the cycle in flight and writes a
// last snapshot.  Without an HDF5 library (snapshot.hpp) snapshots are raw binary files
// outputs/run_<timestamp>/snapshot_%04d.bin instead, and the run says so.
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <filesystem>
#include <stdexcept>
#include <string>

#include "config.hpp"
#include "simulation.hpp"

namespace {
volatile std::sig_atomic_t g_stop = 0;
extern "C" void on_sigint(int) { g_stop = 1; }   // src/main.cpp:13-22

std::string timestamp() {                         // src/utils.cpp: YYYYmmdd_HHMMSS
    char buf[32];
    const std::time_t t = std::time(nullptr);
    std::tm tm{};
    localtime_r(&t, &tm);
    std::strftime(buf, sizeof(buf), "%Y%m%d_%H%M%S", &tm);
    return buf;
}
}  // namespace

int main(int argc, char** argv) {
    using namespace nbh;
    std::signal(SIGINT, on_sigint);
    const std::string config_filename = argc >= 2 ? argv[1] : "simulation.ini";
    Config config;
    try {
        config.load(config_filename);
        if (config.OMEGA_BARYON > config.OMEGA_M)                                    // :36-39
            throw std::runtime_error("OMEGA_BARYON cannot be larger than total OMEGA_M.");
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Error loading %s: %s\n", config_filename.c_str(), e.what());
        return 1;
    }
    std::printf("Successfully loaded %s\n", config_filename.c_str());
    std::printf("N(P,G): (%d^3,%d^3) G: %g fixed_dt: %g\n", config.N_PER_SIDE, config.MESH_SIZE, config.G, config.FIXED_DT);

    const std::string run_dir = "outputs/run_" + timestamp();                        // :53-61
    try {
        std::filesystem::create_directories(run_dir);
        // the reference copies ./simulation.ini whatever file it loaded; the file that was loaded
        // is the one worth keeping
        std::filesystem::copy_file(config_filename, run_dir + "/simulation.ini",
                                   std::filesystem::copy_options::overwrite_existing);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Error preparing %s: %s\n", run_dir.c_str(), e.what());
        return 1;
    }

    try {
        // one process per GPU when a launcher set WORLD_SIZE / RANK / LOCAL_RANK, else GPU 0 alone
        if (std::getenv("WORLD_SIZE")) {
            if (nbmg_init_from_env() != NBMG_OK) throw std::runtime_error(nbmg_last_error());
        } else if (nbpp_init(0) != NBPP_OK) throw std::runtime_error(nbpp_last_error());
        RunOptions options;
        options.log_dir = run_dir;
        options.progress = false;
        if (SnapshotWriter::available()) {
            options.snapshot_target = run_dir;
        } else {
            options.snapshot_target = run_dir + "/snapshot";
            std::printf("no HDF5 library found: snapshots are raw binary (%s_NNNN.bin)\n", options.snapshot_target.c_str());
        }
        Simulation sim(config, options);
        std::printf("output in %s\n", run_dir.c_str());
        const ExitStatus status = sim.run(&g_stop);
        switch (status) {                                                            // :70-84
            case ExitStatus::UserAborted:
                std::printf("\nSimulation aborted by user (Ctrl+C).\n");
                break;
            case ExitStatus::ReachedMaxScaleFactor:
                std::printf("\nSimulation successfully reached target a = %g.\n", config.MAX_SCALE_FACTOR);
                break;
            case ExitStatus::ReachedMaxCycles:
                std::printf("\nSimulation reached maximum allowed cycles (%d)\n", config.MAX_CYCLES);
                break;
        }
        sim.print_timing(status);
        std::printf("%d snapshots written\n", sim.snapshots_written());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "\n[FATAL] Simulation crashed: %s\n", e.what());      // :85-87
        nbgas_shutdown();
        nbpm_shutdown();
        nbmg_shutdown();
        nbpp_shutdown();
        return 1;
    }
    nbgas_shutdown();
    nbpm_shutdown();
    nbmg_shutdown();
    nbpp_shutdown();
    return 0;
}
