#include "config.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <fstream>
#include <map>
#include <stdexcept>

namespace nbh {
namespace {

// Reader for the reference's INI dialect (src/ini.h:25-85): '[section]' headers, 'key = value'
// pairs, a line is a comment only if its first non-blank character is ';', lines without '='
// are ignored, blanks and tabs are trimmed, later duplicates win, missing or unparsable values
// keep the default, booleans accept true/yes/1 and false/no/0 in any case.
class IniFile {
   public:
    explicit IniFile(const std::string& path) {
        std::ifstream in(path);
        if (!in.is_open()) throw std::runtime_error("INI: Could not open file: " + path);
        std::string line, section;
        while (std::getline(in, line)) {
            line = strip(line);
            if (line.empty() || line.front() == ';') continue;
            if (line.front() == '[') {
                section = strip(line.substr(1, line.find(']') - 1));
                continue;
            }
            const auto eq = line.find('=');
            if (eq == std::string::npos) continue;
            values_[section + '\n' + strip(line.substr(0, eq))] = strip(line.substr(eq + 1));
        }
    }

    double number(const char* section, const char* key, double fallback) const {
        const std::string* v = find(section, key);
        if (!v) return fallback;
        try { return std::stod(*v); } catch (...) { return fallback; }
    }
    int integer(const char* section, const char* key, int fallback) const {
        const std::string* v = find(section, key);
        if (!v) return fallback;
        try { return std::stoi(*v); } catch (...) { return fallback; }
    }
    bool flag(const char* section, const char* key, bool fallback) const {
        const std::string* v = find(section, key);
        if (!v) return fallback;
        std::string s = *v;
        std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return std::tolower(c); });
        if (s == "true" || s == "yes" || s == "1") return true;
        if (s == "false" || s == "no" || s == "0") return false;
        return fallback;
    }

   private:
    static std::string strip(const std::string& s) {
        const auto a = s.find_first_not_of(" \t");
        if (a == std::string::npos) return "";
        return s.substr(a, s.find_last_not_of(" \t") - a + 1);
    }
    const std::string* find(const char* section, const char* key) const {
        auto it = values_.find(std::string(section) + '\n' + key);
        return (it == values_.end() || it->second.empty()) ? nullptr : &it->second;
    }
    std::map<std::string, std::string> values_;
};

}  // namespace

Config::Config() { compute_derived_data(); }

// Section and key names: src/config.cpp:14-68.
void Config::load(const std::string& filename) {
    const IniFile ini(filename);
    DOMAIN_SIZE = ini.number("domain", "domain_size", DOMAIN_SIZE);
    MESH_SIZE = ini.integer("domain", "mesh_size", MESH_SIZE);
    BOX_SIZE_MPC = ini.number("domain", "box_size_mpc", BOX_SIZE_MPC);
    OMEGA_BARYON = ini.number("cosmology", "omega_baryon", OMEGA_BARYON);
    OMEGA_M = ini.number("cosmology", "omega_M", OMEGA_M);
    OMEGA_LAMBDA = ini.number("cosmology", "omega_lambda", OMEGA_LAMBDA);
    HUBBLE_PARAM = ini.number("cosmology", "hubble_param", HUBBLE_PARAM);
    EXPANDING_UNIVERSE = ini.flag("cosmology", "expanding_universe", EXPANDING_UNIVERSE);
    N_PER_SIDE = ini.integer("initial_conditions", "n_per_side", N_PER_SIDE);
    STANDING_PARTICLES = ini.flag("initial_conditions", "standing_particles", STANDING_PARTICLES);
    START_A = ini.number("initial_conditions", "start_a", START_A);
    SIGMA_8 = ini.number("initial_conditions", "sigma_8", SIGMA_8);
    SPECTRAL_INDEX = ini.number("initial_conditions", "spectral_index", SPECTRAL_INDEX);
    INITIAL_GAS_TEMPERATURE_K =
        ini.number("initial_conditions", "initial_gas_temp_k", INITIAL_GAS_TEMPERATURE_K);
    SEED = ini.integer("initial_conditions", "seed", SEED);
    USE_HYDRO = ini.flag("hydro", "use_hydro", USE_HYDRO);
    GAMMA = ini.number("hydro", "gamma", GAMMA);
    USE_PM = ini.flag("p3m", "use_pm", USE_PM);
    USE_PP = ini.flag("p3m", "use_pp", USE_PP);
    CUTOFF_RADIUS_CELLS = ini.number("p3m", "cutoff_radius_cells", CUTOFF_RADIUS_CELLS);
    CUTOFF_TRANSITION_WIDTH_FACTOR =
        ini.number("p3m", "cutoff_transition_width_factor", CUTOFF_TRANSITION_WIDTH_FACTOR);
    DT_FACTOR = ini.number("time", "dt_factor", DT_FACTOR);
    MAX_SCALE_FACTOR = ini.number("time", "max_scale_factor", MAX_SCALE_FACTOR);
    CFL_SAFETY_FACTOR = ini.number("time", "cfl_safety_factor", CFL_SAFETY_FACTOR);
    GRAVITY_DT_FACTOR = ini.number("time", "gravity_dt_factor", GRAVITY_DT_FACTOR);
    USE_ADAPTIVE_DT = ini.flag("time", "use_adaptive_dt", USE_ADAPTIVE_DT);
    MAX_CYCLES = ini.integer("time", "max_cycles", MAX_CYCLES);
    SAVE_HDF5_EVERY_DELTA_A = ini.number("output", "save_hdf5_every_delta_a", SAVE_HDF5_EVERY_DELTA_A);
    DEBUG_INFO_EVERY_CYCLES = ini.integer("output", "debug_info_every_cycles", DEBUG_INFO_EVERY_CYCLES);
    ENABLE_ENERGY_DIAGNOSTICS =
        ini.flag("output", "enable_energy_diagnostics", ENABLE_ENERGY_DIAGNOSTICS);
    compute_derived_data();
}

// Derivation order and expressions: src/config.cpp:73-99.
void Config::compute_derived_data() {
    CELL_SIZE = DOMAIN_SIZE / MESH_SIZE;
    CELL_VOLUME = CELL_SIZE * CELL_SIZE * CELL_SIZE;
    OMEGA_DM = OMEGA_M - OMEGA_BARYON;
    CUTOFF_RADIUS = CUTOFF_RADIUS_CELLS * CELL_SIZE;
    CUTOFF_RADIUS_SQUARED = CUTOFF_RADIUS * CUTOFF_RADIUS;
    CUTOFF_TRANSITION_WIDTH = CUTOFF_TRANSITION_WIDTH_FACTOR * CUTOFF_RADIUS;
    R_SWITCH_START = CUTOFF_RADIUS - CUTOFF_TRANSITION_WIDTH;
    R_SWITCH_START_SQ = R_SWITCH_START * R_SWITCH_START;
    NUM_DM_PARTICLES = N_PER_SIDE * N_PER_SIDE * N_PER_SIDE;
    const double total_mass = 1.0;
    G = std::pow(DOMAIN_SIZE, 3) / (6.0 * M_PI * total_mass);
    const double baryon_fraction = OMEGA_BARYON / OMEGA_M;
    DM_PARTICLE_MASS = total_mass * (1.0 - baryon_fraction) / NUM_DM_PARTICLES;
    GAS_TOTAL_MASS = total_mass * baryon_fraction;
    const double spacing = DOMAIN_SIZE / std::cbrt(static_cast<double>(NUM_DM_PARTICLES));
    SOFTENING_SQUARED = std::pow(spacing / 50.0, 2);
    FIXED_DT = DT_FACTOR * (1.0 / std::sqrt(G));
}

nbgas_params Config::gas_params() const {
    nbgas_params p;
    p.cell_size = CELL_SIZE;
    p.gamma = GAMMA;
    p.cfl_safety_factor = CFL_SAFETY_FACTOR;
    p.mesh_size = MESH_SIZE;
    p.reserved = 0;
    return p;
}

nbpm_params Config::mesh_params() const {
    nbpm_params p;
    p.domain_size = DOMAIN_SIZE;
    p.G = G;
    p.mesh_size = MESH_SIZE;
    p.reserved = 0;
    return p;
}

nbpp_params Config::pp_params() const {
    nbpp_params p;
    p.domain_size = DOMAIN_SIZE;
    p.G = G;
    p.soft_sq = SOFTENING_SQUARED;
    p.pm_soft_sq = (0.5 * CELL_SIZE) * (0.5 * CELL_SIZE);
    p.r_switch = R_SWITCH_START;
    p.r_switch_sq = R_SWITCH_START_SQ;
    p.cutoff_sq = CUTOFF_RADIUS_SQUARED;
    p.trans_width = CUTOFF_TRANSITION_WIDTH;
    p.cell_size = CELL_SIZE;
    p.mesh_size = MESH_SIZE;
    p.use_pm = USE_PM ? 1 : 0;
    return p;
}

}  // namespace nbh
