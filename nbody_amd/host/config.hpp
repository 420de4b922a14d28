// config.hpp - the engine's run configuration, restating the reference's Config
// (src/config.h:4-55, src/config.cpp:8-99) for the fields the particle-particle path and its
// callers read.  Same member names, defaults and derivation order as the reference, so that a
// simulation.ini written for it loads unchanged.
#pragma once
#include <string>

#include "../../include/nbgas.h"
#include "../../include/nbpm.h"
#include "../../include/nbpp.h"

namespace nbh {

struct Config {
    // [domain]                                   src/config.h:5-7
    double DOMAIN_SIZE = 1.0;
    int MESH_SIZE = 32;
    double BOX_SIZE_MPC = 16.0;
    // [cosmology]                                src/config.h:9-13
    double OMEGA_BARYON = 0.045;
    double OMEGA_M = 0.3;
    double OMEGA_LAMBDA = 0.7;
    double HUBBLE_PARAM = 0.7;
    bool EXPANDING_UNIVERSE = true;
    // [initial_conditions]                       src/config.h:15-21
    int N_PER_SIDE = 32;
    bool STANDING_PARTICLES = false;
    double START_A = 0.02;
    double SIGMA_8 = 0.81;
    double SPECTRAL_INDEX = 0.96;
    double INITIAL_GAS_TEMPERATURE_K = 50.0;
    int SEED = 42;
    // [hydro]                                    src/config.h:23-24
    bool USE_HYDRO = true;
    double GAMMA = 5.0 / 3.0;
    // [p3m]                                      src/config.h:26-29
    bool USE_PM = true;
    bool USE_PP = true;
    double CUTOFF_RADIUS_CELLS = 2.5;
    double CUTOFF_TRANSITION_WIDTH_FACTOR = 0.2;
    // [time]                                     src/config.h:31-36
    double DT_FACTOR = 1e-3;
    double MAX_SCALE_FACTOR = 1.0;
    double CFL_SAFETY_FACTOR = 0.5;
    double GRAVITY_DT_FACTOR = 0.2;
    bool USE_ADAPTIVE_DT = true;
    int MAX_CYCLES = 1000000000;
    // [output]                                   src/config.h:38-40
    double SAVE_HDF5_EVERY_DELTA_A = 0.005;
    int DEBUG_INFO_EVERY_CYCLES = 40;
    bool ENABLE_ENERGY_DIAGNOSTICS = true;

    // derived                                    src/config.h:43-50
    double G = 0.0;
    double CELL_SIZE = 0.0, CELL_VOLUME = 0.0, OMEGA_DM = 0.0;
    double CUTOFF_RADIUS = 0.0, CUTOFF_RADIUS_SQUARED = 0.0;
    double CUTOFF_TRANSITION_WIDTH = 0.0, R_SWITCH_START = 0.0, R_SWITCH_START_SQ = 0.0;
    int NUM_DM_PARTICLES = 0;
    double DM_PARTICLE_MASS = 0.0, GAS_TOTAL_MASS = 0.0;
    double SOFTENING_SQUARED = 0.0;
    double FIXED_DT = 0.0;

    Config();                                  // defaults + compute_derived_data (config.cpp:8)
    void load(const std::string& filename);    // throws std::runtime_error if unreadable
    void compute_derived_data();               // config.cpp:73-99

    // The scalars compute_pp_forces hoists (src/particles.cpp:143-152) as the library's POD.
    nbpp_params pp_params() const;
    // What the mesh solve reads (src/integrator.cpp:48, :91, :100).
    nbpm_params mesh_params() const;
    // What the gas solver reads (src/gas.cpp:191, :251, :322).
    nbgas_params gas_params() const;
};

}  // namespace nbh
