// config.hpp - the engine's run configuration, restating the reference's Config
// (src/config.h:4-55, src/config.cpp:8-99) for the fields the particle-particle path and its
// callers read.  Same member names, defaults and derivation order as the reference, so that a
// simulation.ini written for it loads unchanged.
#pragma once
#include <string>

#include "../../include/nbgas.h"
#include "../../include/nbmg.h"
#include "../../include/nbpm.h"
#include "../../include/nbpp.h"

namespace nbh {

struct Config {
    // Every run-time field keeps the reference's name and default (src/config.h:5-40), grouped here by
    // the part of the step that reads it; the comment gives the simulation.ini key, as [section] key.
    // -- the short-range (particle-particle) force and its cutoff
    bool USE_PM{true};                              // [p3m] use_pm: mesh (long-range) force
    bool USE_PP{true};                              // [p3m] use_pp: particle-particle (short-range) force
    double CUTOFF_RADIUS_CELLS{2.5};                // [p3m] cutoff_radius_cells: r_c in mesh cells
    double CUTOFF_TRANSITION_WIDTH_FACTOR{0.2};     // [p3m] cutoff_transition_width_factor: taper width / r_c

    // -- box and mesh
    double DOMAIN_SIZE{1.0};                        // [domain] domain_size: box length, code units
    int MESH_SIZE{32};                              // [domain] mesh_size: PM / gas cells per side
    double BOX_SIZE_MPC{16.0};                      // [domain] box_size_mpc: the same length in Mpc

    // -- time stepping
    double DT_FACTOR{1e-3};                         // [time] dt_factor: FIXED_DT in units of 1 / sqrt(G)
    double MAX_SCALE_FACTOR{1.0};                   // [time] max_scale_factor: the run ends there
    double CFL_SAFETY_FACTOR{0.5};                  // [time] cfl_safety_factor
    double GRAVITY_DT_FACTOR{0.2};                  // [time] gravity_dt_factor
    bool USE_ADAPTIVE_DT{true};                     // [time] use_adaptive_dt: min(CFL, gravity, FIXED_DT)
    int MAX_CYCLES{1000000000};                     // [time] max_cycles

    // -- initial conditions
    int N_PER_SIDE{32};                             // [initial_conditions] n_per_side: particles per side
    bool STANDING_PARTICLES{false};                 // [initial_conditions] standing_particles: dark matter frozen
    double START_A{0.02};                           // [initial_conditions] start_a: initial scale factor
    double SIGMA_8{0.81};                           // [initial_conditions] sigma_8
    double SPECTRAL_INDEX{0.96};                    // [initial_conditions] spectral_index: n_s
    double INITIAL_GAS_TEMPERATURE_K{50.0};         // [initial_conditions] initial_gas_temp_k
    int SEED{42};                                   // [initial_conditions] seed of the white-noise field

    // -- background cosmology
    double OMEGA_BARYON{0.045};                     // [cosmology] omega_baryon
    double OMEGA_M{0.3};                            // [cosmology] omega_M (baryons included)
    double OMEGA_LAMBDA{0.7};                       // [cosmology] omega_lambda
    double HUBBLE_PARAM{0.7};                       // [cosmology] hubble_param: h
    bool EXPANDING_UNIVERSE{true};                  // [cosmology] expanding_universe: false = static box, a = 1

    // -- gas
    bool USE_HYDRO{true};                           // [hydro] use_hydro
    double GAMMA{5.0 / 3.0};                        // [hydro] gamma: adiabatic index

    // -- snapshots and diagnostics
    double SAVE_HDF5_EVERY_DELTA_A{0.005};          // [output] save_hdf5_every_delta_a: snapshot cadence in a
    int DEBUG_INFO_EVERY_CYCLES{40};                // [output] debug_info_every_cycles: diagnostics cadence
    bool ENABLE_ENERGY_DIAGNOSTICS{true};           // [output] enable_energy_diagnostics

    // Derived by compute_derived_data (src/config.h:43-50, src/config.cpp:73-99); total mass is 1:
    double CELL_SIZE{0.0};                          // DOMAIN_SIZE / MESH_SIZE
    double CELL_VOLUME{0.0};                        // CELL_SIZE^3
    double OMEGA_DM{0.0};                           // OMEGA_M - OMEGA_BARYON
    double G{0.0};                                  // DOMAIN_SIZE^3 / (6 pi): the constant of these units
    double CUTOFF_RADIUS{0.0};                      // CUTOFF_RADIUS_CELLS * CELL_SIZE
    double CUTOFF_RADIUS_SQUARED{0.0};
    double CUTOFF_TRANSITION_WIDTH{0.0};            // CUTOFF_TRANSITION_WIDTH_FACTOR * CUTOFF_RADIUS
    double R_SWITCH_START{0.0};                     // CUTOFF_RADIUS - CUTOFF_TRANSITION_WIDTH: the taper begins
    double R_SWITCH_START_SQ{0.0};
    int NUM_DM_PARTICLES{0};                        // N_PER_SIDE^3
    double DM_PARTICLE_MASS{0.0};                   // (1 - OMEGA_BARYON / OMEGA_M) / NUM_DM_PARTICLES
    double GAS_TOTAL_MASS{0.0};                     // OMEGA_BARYON / OMEGA_M
    double SOFTENING_SQUARED{0.0};                  // (mean particle spacing / 50)^2
    double FIXED_DT{0.0};                           // DT_FACTOR / sqrt(G)

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
