// snapshot.hpp - snapshot files with the reference's HDF5 layout (src/hdf5_writter.cpp:57-134):
// root attributes :68-99, /particles/{position,velocity,acceleration}_{x,y,z} and mass :101-116,
// /gas/{density,momentum_x,momentum_y,momentum_z,energy,pressure} as mesh^3 datasets :118-127,
// file name snapshot_%04d.hdf5 in the run directory :59-61 - what scripts/validate_run.py and
// scripts/viewer.py open.
//
// The HDF5 C library is bound at run time (dlopen), not at link time: the engine and the three
// GPU libraries build and run on a machine without it, and SnapshotWriter's constructor throws
// there.  Search order: $NBODY_HDF5_LIB, libhdf5.so on the loader path, /opt/conda/lib/libhdf5.so.
#pragma once
#include <string>
#include <vector>

#include "config.hpp"
#include "particles.hpp"

namespace nbh {

class SnapshotWriter {
public:
    SnapshotWriter(const std::string& run_dir, const Config& config);   // throws without an HDF5 library
    ~SnapshotWriter();
    SnapshotWriter(const SnapshotWriter&) = delete;
    SnapshotWriter& operator=(const SnapshotWriter&) = delete;

    // `gas` holds the nine meshes in nbgas_state order (density, momentum x y z, energy, pressure,
    // velocity x y z); the first six are written when config.USE_HYDRO is set.  Throws
    // std::runtime_error if the file cannot be written (the reference prints and carries on,
    // :130-133; a run that silently loses its output is not worth its GPU time).
    void save_snapshot(int snapshot_index, long cycle_count, const ParticleSystem& dm,
                       const std::vector<double> (&gas)[9], double total_time, double scale_factor,
                       const Config& config);

    static bool available();   // an HDF5 library can be loaded
    std::string path_of(int snapshot_index) const;

private:
    struct Api;
    Api* api_;
    std::string output_directory_;
};

}  // namespace nbh
