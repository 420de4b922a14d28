// snapshot.cpp - see snapshot.hpp.  HDF5 1.10+ C API (hid_t is 64-bit), resolved with dlsym.
#include "snapshot.hpp"

#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

namespace nbh {

namespace {
using hid = std::int64_t;
using herr = int;
using hsize = unsigned long long;
constexpr unsigned ACC_TRUNC = 0x0002u;   // H5F_ACC_TRUNC
constexpr unsigned ACC_SWMR_WRITE = 0x0020u;   // H5F_ACC_SWMR_WRITE
constexpr hid DEFAULT = 0;                // H5P_DEFAULT, H5S_ALL
constexpr int CLASS_SCALAR = 0;           // H5S_SCALAR

void* open_library() {
    if (const char* env = std::getenv("NBODY_HDF5_LIB"))
        if (void* h = dlopen(env, RTLD_NOW | RTLD_LOCAL)) return h;
    for (const char* name : {"libhdf5.so", "libhdf5_serial.so", "/opt/conda/lib/libhdf5.so"})
        if (void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) return h;
    return nullptr;
}
}  // namespace

struct SnapshotWriter::Api {
    void* handle = nullptr;
    herr (*open)() = nullptr;
    herr (*get_libversion)(unsigned*, unsigned*, unsigned*) = nullptr;
    hid (*fcreate)(const char*, unsigned, hid, hid) = nullptr;
    herr (*fclose)(hid) = nullptr;
    hid (*gcreate)(hid, const char*, hid, hid, hid) = nullptr;
    herr (*gclose)(hid) = nullptr;
    hid (*screate)(int) = nullptr;
    hid (*screate_simple)(int, const hsize*, const hsize*) = nullptr;
    herr (*sclose)(hid) = nullptr;
    hid (*dcreate)(hid, const char*, hid, hid, hid, hid, hid) = nullptr;
    herr (*dwrite)(hid, hid, hid, hid, hid, const void*) = nullptr;
    herr (*dclose)(hid) = nullptr;
    hid (*acreate)(hid, const char*, hid, hid, hid, hid) = nullptr;
    herr (*awrite)(hid, hid, const void*) = nullptr;
    herr (*aclose)(hid) = nullptr;
    hid native_double = -1, native_int = -1;

    template <class F>
    void bind(F& f, const char* name) {
        f = reinterpret_cast<F>(dlsym(handle, name));
        if (!f) throw std::runtime_error(std::string("HDF5 library lacks ") + name);
    }

    Api() {
        handle = open_library();
        if (!handle)
            throw std::runtime_error("no HDF5 library found (set NBODY_HDF5_LIB to a libhdf5.so, version 1.10 or later)");
        bind(open, "H5open");
        bind(get_libversion, "H5get_libversion");
        bind(fcreate, "H5Fcreate");
        bind(fclose, "H5Fclose");
        bind(gcreate, "H5Gcreate2");
        bind(gclose, "H5Gclose");
        bind(screate, "H5Screate");
        bind(screate_simple, "H5Screate_simple");
        bind(sclose, "H5Sclose");
        bind(dcreate, "H5Dcreate2");
        bind(dwrite, "H5Dwrite");
        bind(dclose, "H5Dclose");
        bind(acreate, "H5Acreate2");
        bind(awrite, "H5Awrite");
        bind(aclose, "H5Aclose");
        unsigned maj = 0, min = 0, rel = 0;
        if (open() < 0 || get_libversion(&maj, &min, &rel) < 0) throw std::runtime_error("H5open failed");
        if (maj == 1 && min < 10)   // 1.8 has 32-bit identifiers: the signatures above would not match
            throw std::runtime_error("HDF5 " + std::to_string(maj) + "." + std::to_string(min) + " is older than 1.10");
        auto* d = static_cast<hid*>(dlsym(handle, "H5T_NATIVE_DOUBLE_g"));
        auto* i = static_cast<hid*>(dlsym(handle, "H5T_NATIVE_INT_g"));
        if (!d || !i || *d < 0 || *i < 0) throw std::runtime_error("HDF5 native type identifiers not found");
        native_double = *d;
        native_int = *i;
    }
    ~Api() {
        if (handle) dlclose(handle);
    }
};

namespace {
void check(std::int64_t rc, const char* what, const std::string& file) {
    if (rc < 0) throw std::runtime_error(std::string("HDF5: ") + what + " failed for " + file);
}
}  // namespace

bool SnapshotWriter::available() {
    void* h = open_library();
    if (h) dlclose(h);
    return h != nullptr;
}

SnapshotWriter::SnapshotWriter(const std::string& run_dir, const Config&) : api_(new Api), output_directory_(run_dir) {}

SnapshotWriter::~SnapshotWriter() { delete api_; }

std::string SnapshotWriter::path_of(int snapshot_index) const {
    char name[64];
    std::snprintf(name, sizeof(name), "/snapshot_%04d.hdf5", snapshot_index);   // src/hdf5_writter.cpp:60
    return output_directory_ + name;
}

void SnapshotWriter::save_snapshot(int snapshot_index, long /*cycle_count*/, const ParticleSystem& dm,
                                   const std::vector<double> (&gas)[9], double total_time, double scale_factor,
                                   const Config& config) {
    const Api& h = *api_;
    const std::string file = path_of(snapshot_index);
    // Single-writer / multiple-reader access, as the reference opens its files (:64): a viewer may
    // poll the directory and read a snapshot while the run is still writing.  A library that
    // refuses the flag (before 1.10) gets the plain file.
    hid f = h.fcreate(file.c_str(), ACC_TRUNC | ACC_SWMR_WRITE, DEFAULT, DEFAULT);
    if (f < 0) f = h.fcreate(file.c_str(), ACC_TRUNC, DEFAULT, DEFAULT);
    check(f, "H5Fcreate", file);
    const hid scalar = h.screate(CLASS_SCALAR);
    check(scalar, "H5Screate", file);
    auto attr_double = [&](const char* name, double v) {            // :5-12
        const hid a = h.acreate(f, name, h.native_double, scalar, DEFAULT, DEFAULT);
        check(a, "H5Acreate2", file);
        check(h.awrite(a, h.native_double, &v), "H5Awrite", file);
        h.aclose(a);
    };
    auto attr_int = [&](const char* name, int v) {                  // :14-27 (flags are stored as int 0/1)
        const hid a = h.acreate(f, name, h.native_int, scalar, DEFAULT, DEFAULT);
        check(a, "H5Acreate2", file);
        check(h.awrite(a, h.native_int, &v), "H5Awrite", file);
        h.aclose(a);
    };
    auto dataset = [&](hid group, const char* name, const std::vector<double>& v, int rank, const hsize* dims) {
        const hid space = h.screate_simple(rank, dims, nullptr);
        check(space, "H5Screate_simple", file);
        const hid d = h.dcreate(group, name, h.native_double, space, DEFAULT, DEFAULT, DEFAULT);
        check(d, "H5Dcreate2", file);
        check(h.dwrite(d, h.native_double, DEFAULT, DEFAULT, DEFAULT, v.data()), "H5Dwrite", file);
        h.dclose(d);
        h.sclose(space);
    };
    // attributes on the file identifier land on the root group, in the reference's order (:68-99)
    attr_double("domain_size", config.DOMAIN_SIZE);
    attr_double("box_size_mpc", config.BOX_SIZE_MPC);
    attr_int("mesh_size", config.MESH_SIZE);
    attr_double("omega_baryon", config.OMEGA_BARYON);
    attr_double("omega_M", config.OMEGA_M);
    attr_double("omega_lambda", config.OMEGA_LAMBDA);
    attr_double("hubble_param", config.HUBBLE_PARAM);
    attr_int("n_per_side", config.N_PER_SIDE);
    attr_int("use_hydro", config.USE_HYDRO);
    attr_double("g_const", config.G);
    attr_double("gamma", config.GAMMA);
    attr_int("standing_particles", config.STANDING_PARTICLES);
    attr_int("expanding_universe", config.EXPANDING_UNIVERSE);
    attr_double("initial_scale_factor", config.START_A);
    attr_double("primordial_index", config.SPECTRAL_INDEX);
    attr_double("sigma_8", config.SIGMA_8);
    attr_int("use_pm", config.USE_PM);
    attr_int("use_pp", config.USE_PP);
    attr_double("cutoff_radius_cells", config.CUTOFF_RADIUS_CELLS);
    attr_double("dt_factor", config.DT_FACTOR);
    attr_double("cfl_safety_factor", config.CFL_SAFETY_FACTOR);
    attr_double("gravity_dt_factor", config.GRAVITY_DT_FACTOR);
    attr_int("use_adaptive_dt", config.USE_ADAPTIVE_DT);
    attr_int("seed", config.SEED);
    attr_double("simulation_time", total_time);
    attr_double("scale_factor", scale_factor);

    const hid particles = h.gcreate(f, "particles", DEFAULT, DEFAULT, DEFAULT);   // :101-116
    check(particles, "H5Gcreate2", file);
    const hsize n[1] = {static_cast<hsize>(dm.num_particles)};
    const std::pair<const char*, const std::vector<double>*> columns[] = {
        {"position_x", &dm.pos_x}, {"position_y", &dm.pos_y}, {"position_z", &dm.pos_z},
        {"velocity_x", &dm.vel_x}, {"velocity_y", &dm.vel_y}, {"velocity_z", &dm.vel_z},
        {"acceleration_x", &dm.acc_x}, {"acceleration_y", &dm.acc_y}, {"acceleration_z", &dm.acc_z},
        {"mass", &dm.mass}};
    for (const auto& c : columns) {
        if (c.second->size() != n[0]) throw std::runtime_error(std::string("snapshot: wrong length of ") + c.first);
        dataset(particles, c.first, *c.second, 1, n);
    }
    h.gclose(particles);

    if (config.USE_HYDRO) {                                                         // :118-127
        const hid group = h.gcreate(f, "gas", DEFAULT, DEFAULT, DEFAULT);
        check(group, "H5Gcreate2", file);
        const hsize m = static_cast<hsize>(config.MESH_SIZE);
        const hsize dims[3] = {m, m, m};
        const char* names[6] = {"density", "momentum_x", "momentum_y", "momentum_z", "energy", "pressure"};
        for (int k = 0; k < 6; ++k) {
            if (gas[k].size() != m * m * m) throw std::runtime_error(std::string("snapshot: wrong size of gas ") + names[k]);
            dataset(group, names[k], gas[k], 3, dims);
        }
        h.gclose(group);
    }
    h.sclose(scalar);
    check(h.fclose(f), "H5Fclose", file);
}

}  // namespace nbh
