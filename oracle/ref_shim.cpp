// ref_shim.cpp - extern "C" doorway onto the reference's OWN config.cpp and math_utils.h.
//
// TEST INFRASTRUCTURE ONLY.  This file holds no algorithm: it includes the reference's
// headers where they lie (-I/root/reference/src) and is linked with the reference's
// unmodified src/config.cpp into oracle/_ref/libnbref.so (see oracle/Makefile).  It is
// used to pin orc_config_derive / orc_periodic_displacement and to generate
// tests/golden/config_derived.json.  The rest of the reference's particle path
// (src/particles.cpp) needs <Eigen/Dense>, which this image lacks, so it is not built.
#include <cstring>
#include <string>

#include "config.h"      // reference src/config.h
#include "math_utils.h"  // reference src/math_utils.h

extern "C" {

double ref_periodic_displacement(double dx, double domain_size) {
    return periodic_displacement(dx, domain_size);
}

struct ref_config_view {
    double domain_size;
    int mesh_size, n_per_side;
    double omega_baryon, omega_m, cutoff_radius_cells, cutoff_transition_width_factor,
        dt_factor, gravity_dt_factor;
    int use_pm, use_pp, standing_particles, seed;
    // derived
    double cell_size, cell_volume, omega_dm, cutoff_radius, cutoff_radius_squared,
        cutoff_transition_width, r_switch_start, r_switch_start_sq;
    int num_dm_particles, pad;
    double G, dm_particle_mass, gas_total_mass, softening_squared, fixed_dt;
};

static void view_of(const Config& c, ref_config_view* v) {
    v->domain_size = c.DOMAIN_SIZE;
    v->mesh_size = c.MESH_SIZE;
    v->n_per_side = c.N_PER_SIDE;
    v->omega_baryon = c.OMEGA_BARYON;
    v->omega_m = c.OMEGA_M;
    v->cutoff_radius_cells = c.CUTOFF_RADIUS_CELLS;
    v->cutoff_transition_width_factor = c.CUTOFF_TRANSITION_WIDTH_FACTOR;
    v->dt_factor = c.DT_FACTOR;
    v->gravity_dt_factor = c.GRAVITY_DT_FACTOR;
    v->use_pm = c.USE_PM;
    v->use_pp = c.USE_PP;
    v->standing_particles = c.STANDING_PARTICLES;
    v->seed = c.SEED;
    v->cell_size = c.CELL_SIZE;
    v->cell_volume = c.CELL_VOLUME;
    v->omega_dm = c.OMEGA_DM;
    v->cutoff_radius = c.CUTOFF_RADIUS;
    v->cutoff_radius_squared = c.CUTOFF_RADIUS_SQUARED;
    v->cutoff_transition_width = c.CUTOFF_TRANSITION_WIDTH;
    v->r_switch_start = c.R_SWITCH_START;
    v->r_switch_start_sq = c.R_SWITCH_START_SQ;
    v->num_dm_particles = c.NUM_DM_PARTICLES;
    v->pad = 0;
    v->G = c.G;
    v->dm_particle_mass = c.DM_PARTICLE_MASS;
    v->gas_total_mass = c.GAS_TOTAL_MASS;
    v->softening_squared = c.SOFTENING_SQUARED;
    v->fixed_dt = c.FIXED_DT;
}

// Struct defaults (src/config.h:5-40) + compute_derived_data.
void ref_config_defaults(ref_config_view* v) {
    Config c;
    view_of(c, v);
}

// Config::load of an ini file (src/config.cpp:10-71).  Returns 0, or -1 if load threw.
int ref_config_load(const char* path, ref_config_view* v) {
    try {
        Config c;
        c.load(std::string(path));
        view_of(c, v);
        return 0;
    } catch (...) {
        return -1;
    }
}

// Overwrite the independent fields, then compute_derived_data (src/config.cpp:73-99).
void ref_config_derive(double domain_size, int mesh_size, int n_per_side, double omega_baryon,
                       double omega_m, double cutoff_radius_cells,
                       double cutoff_transition_width_factor, double dt_factor,
                       ref_config_view* v) {
    Config c;
    c.DOMAIN_SIZE = domain_size;
    c.MESH_SIZE = mesh_size;
    c.N_PER_SIDE = n_per_side;
    c.OMEGA_BARYON = omega_baryon;
    c.OMEGA_M = omega_m;
    c.CUTOFF_RADIUS_CELLS = cutoff_radius_cells;
    c.CUTOFF_TRANSITION_WIDTH_FACTOR = cutoff_transition_width_factor;
    c.DT_FACTOR = dt_factor;
    c.compute_derived_data();
    view_of(c, v);
}
}
