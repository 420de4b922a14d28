"""Regenerates the committed golden vectors.  Run in the build container only:

    python tests/golden/make_golden.py

Sources of truth, in order of authority:
  * reference_config.json, reference_wrap.json: outputs of the reference's OWN compiled
    src/config.cpp and src/math_utils.h (oracle/_ref/libnbref.so, built by oracle/Makefile
    from the files under /root/reference).
  * reference_kat.json: the inputs and expected values written in the reference's own
    tests (tests/test_gravity.cpp, tests/test_math.cpp:10-48, tests/test_hydro.cpp:29-67,
    tests/test_ics.cpp), transcribed as data, plus two-body cases whose expected values the
    tests derive from the reference's written specification (docs/source/book.md:384-455).
  * pp_small.npz: small particle sets with accelerations from the oracle's independent
    long-double direct sum (oracle/pp_oracle.c: orc_pp_accumulate_ld).  The reference's
    particles.cpp cannot be built in this image (no Eigen), so these are oracle outputs,
    pinned by the two files above.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.oracle import Oracle, Reference  # noqa: E402
import datasets  # noqa: E402


def main():
    ref = Reference()
    orc = Oracle()

    cfg = {"defaults": ref.config_defaults(),
           "simulation_ini": ref.config_load("/root/reference/simulation.ini")}
    for name, kw in {
        "pp128": dict(mesh_size=128, n_per_side=128),
        "pp256": dict(mesh_size=256, n_per_side=256),
        "odd31": dict(mesh_size=31, n_per_side=32),
        "box64": dict(domain_size=64.0, mesh_size=16, n_per_side=16),
        "box10_wide": dict(domain_size=10.0, mesh_size=20, n_per_side=12,
                           cutoff_radius_cells=3.0, cutoff_transition_width_factor=0.35),
        "test_gravity_scaled": dict(mesh_size=4, n_per_side=2),  # tests/test_gravity.cpp:237-242
    }.items():
        cfg[name] = ref.config_derive(**kw)
    with open(os.path.join(HERE, "reference_config.json"), "w") as f:
        json.dump(cfg, f, indent=1, sort_keys=True)

    rng = np.random.Generator(np.random.MT19937(99))
    wrap = []
    for L in (64.0, 1.0, 10.0, 0.3):
        h = 0.5 * L
        probes = [62.0 / 64 * L, -62.0 / 64 * L, -2.0 / 64 * L, h, -h, np.nextafter(h, 2 * L),
                  np.nextafter(-h, -2 * L), np.nextafter(h, 0), 0.0, -0.0, 0.999 * L, -0.999 * L,
                  1.4 * L, -1.4 * L] + list((rng.random(40) * 2 - 1) * L)
        for d in probes:
            wrap.append({"dx": float(d).hex(), "L": float(L).hex(),
                         "out": float(ref.periodic_displacement(d, L)).hex()})
    with open(os.path.join(HERE, "reference_wrap.json"), "w") as f:
        json.dump(wrap, f, indent=0)

    kat = {
        "two_body": {  # tests/test_gravity.cpp:100-113
            "source": "tests/test_gravity.cpp:98-155",
            "config": {"MESH_SIZE": 4, "CELL_SIZE": 2.0, "DOMAIN_SIZE": 8.0, "G": 1.0,
                       "SOFTENING_SQUARED": 0.0},
            "particles": [[2.0, 2.0, 2.0, 1.0], [5.0, 2.0, 2.0, 1.0]],
            "newtonian": {"USE_PM": False, "CUTOFF_RADIUS_SQUARED": 16.0,
                          "acc_x": [1.0 / 9.0, -1.0 / 9.0], "acc_y0": 0.0, "acc_z0": 0.0},
            "p3m_cutoff": {"USE_PM": True, "CUTOFF_RADIUS": 2.0, "CUTOFF_RADIUS_SQUARED": 4.0,
                           "R_SWITCH_START": 1.0, "R_SWITCH_START_SQ": 1.0,
                           "CUTOFF_TRANSITION_WIDTH": 1.0, "acc_x": [0.0, 0.0]},
            "approx_rel": 1.2e-5,  # Catch::Approx default epsilon, scale = value
        },
        "periodic_displacement": {  # tests/test_math.cpp:10-48
            "source": "tests/test_math.cpp:10-48", "box_size": 64.0,
            "cases": [{"dx": 62.0, "expect": -2.0}, {"dx": -62.0, "expect": 2.0},
                      {"dx": -2.0, "expect": -2.0}, {"dx": 32.0, "expect_abs": 32.0}],
        },
    }
    # Values a build of the reference printed for the shipped simulation.ini (seed 42, 32^3,
    # mesh 32) during the survey of the reference; transcribed, not regenerated here.
    kat["default_ini_anchor"] = {
        "source": "initialize_state (src/ics.cpp:290-349) on /root/reference/simulation.ini",
        "scale_factor": 0.020000000000000007, "hubble_param": 235.70446027301335,
        "p0_pos": [0.016061997996275945, 0.012630329609784496, 0.014442457643764284],
        "p0_vel": [0.10300131937684433, -0.70584992132590851, -0.27872764623469792],
        "p0_acc_initial_pm_plus_pp": [0.00038592991125482439, -0.0018989432424388109,
                                      -0.00086020514888652676],
        "max_accel_sq": 2.5834272123748411e-05,
        "p0_acc_after_one_step": [0.00066809247978902883, -0.0033187059537092838,
                                  -0.0015552108792028102],
        "after_one_step_source": "KDK_step (src/integrator.cpp:215-275) with dt = FIXED_DT on the "
                                 "state of initialize_state, /root/reference/simulation.ini "
                                 "(mesh + short range + gas)",
    }
    kat["hll_solver"] = {  # tests/test_hydro.cpp:29-67; S_R_minus_S_L is only read in the mixed regime
        "source": "tests/test_hydro.cpp:29-67",
        "cases": [
            {"S_L": 1.5, "S_R": 3.0, "S_R_minus_S_L": 0.0, "FL": 10.0, "FR": 20.0, "UL": 1.0,
             "UR": 2.0, "flux": 10.0},
            {"S_L": -3.0, "S_R": -1.5, "S_R_minus_S_L": 0.0, "FL": 10.0, "FR": 20.0, "UL": 1.0,
             "UR": 2.0, "flux": 20.0},
            {"S_L": -1.0, "S_R": 2.0, "S_R_minus_S_L": 3.0, "FL": 4.0, "FR": 10.0, "UL": 2.0,
             "UR": 5.0, "flux": 4.0},
        ]}
    # ---- the rest of what the reference's own tests hold for the gravity step, as data ----
    V = 8.0
    kat["cic_deposit"] = {  # tests/test_gravity.cpp:13-93
        "source": "tests/test_gravity.cpp:13-93",
        "config": {"MESH_SIZE": 4, "CELL_SIZE": 2.0, "CELL_VOLUME": V, "DOMAIN_SIZE": 8.0},
        "cases": [
            {"name": "on a grid vertex", "particle": [2.0, 4.0, 6.0, 16.0], "total_mass": 16.0,
             "cells": [[1, 2, 3, 16.0 / V], [2, 2, 3, 0.0]]},
            {"name": "dead centre of a cell", "particle": [3.0, 3.0, 3.0, 8.0], "total_mass": 8.0,
             "cells": [[i, j, k, 1.0 / V] for k in (1, 2) for j in (1, 2) for i in (1, 2)]},
            {"name": "on the periodic boundary", "particle": [7.0, 0.0, 0.0, 10.0], "total_mass": 10.0,
             "cells": [[3, 0, 0, 5.0 / V], [0, 0, 0, 5.0 / V]]},
        ],
        "approx_rel": 1.2e-5}
    kat["pm_solver"] = {  # tests/test_gravity.cpp:160-206
        "source": "tests/test_gravity.cpp:160-206",
        "config": {"MESH_SIZE": 4, "CELL_SIZE": 2.0, "DOMAIN_SIZE": 8.0, "G": 1.0},
        "uniform": {"density": 5.0, "margin": 1e-10},
        "point_mass": {"particle": [2.0, 2.0, 2.0, 100.0], "cell": [1, 1, 1], "margin": 1e-10,
                       "right_neighbour_negative": [2, 1, 1], "left_neighbour_positive": [0, 1, 1]}}
    kat["cic_gather"] = {  # tests/test_gravity.cpp:208-233
        "source": "tests/test_gravity.cpp:208-233",
        "config": {"MESH_SIZE": 4, "CELL_SIZE": 2.0, "DOMAIN_SIZE": 8.0},
        "particle": [3.0, 3.0, 3.0, 5.0], "uniform_field": [2.0, -1.5, 0.0]}
    kat["pm_plus_pp_active"] = {  # tests/test_gravity.cpp:235-269
        "source": "tests/test_gravity.cpp:235-269",
        "config": {"MESH_SIZE": 4, "N_PER_SIDE": 2, "START_A": 0.5},
        "particles": [[0.5, 0.5, 0.4, 0.5], [0.5, 0.5, 0.6, 0.5]],
        "min_abs_acc_z0": 0.05}
    kat["ics"] = {  # tests/test_ics.cpp:7-299
        "source": "tests/test_ics.cpp:7-299",
        "internal_energy": {"T_kelvin": 50.0, "hubble_param": 0.7, "box_size_mpc": 100.0, "domain_size": 1.0,
                            "gamma": 5.0 / 3.0, "u_code": 4.60265e-9, "margin": 1e-13,
                            "double_domain_factor": 4.0},
        "macro_state": {"config": {"MESH_SIZE": 32, "N_PER_SIDE": 32, "BOX_SIZE_MPC": 100.0, "USE_PP": False},
                        "total_mass": 1.0, "mass_margin": 1e-6, "dm_momentum_x_below": 1e-10,
                        "gas_momentum_x_below": 1e-3, "max_abs_vel_x_between": [0.1, 0.6]},
        "sigma8_linearity": {"config": {"MESH_SIZE": 32, "BOX_SIZE_MPC": 100.0, "SEED": 42, "SPECTRAL_INDEX": 0.96},
                             "start_a": 0.02, "sigma_8": [0.8, 1.6], "factor": 2.0, "margin": 1e-12},
        "zeldovich_kinematics": {"config": {"MESH_SIZE": 16, "N_PER_SIDE": 16, "BOX_SIZE_MPC": 100.0,
                                            "OMEGA_BARYON": 0.048, "SIGMA_8": 0.81, "SEED": 123},
                                 "lattice_margin": 1e-7},
        "gas_traces_dm": {"config": {"MESH_SIZE": 16, "N_PER_SIDE": 16, "BOX_SIZE_MPC": 100.0,
                                     "OMEGA_BARYON": 0.048, "SIGMA_8": 0.81, "SEED": 42},
                          "floor": 1e-12, "margin": 1e-10},
        "zero_mean": {"config": {"MESH_SIZE": 32, "BOX_SIZE_MPC": 100.0, "SIGMA_8": 0.81, "SEED": 42},
                      "start_a": 0.02, "mean_below": 1e-14},
        "standing_particles": {"config": {"MESH_SIZE": 16, "N_PER_SIDE": 16, "STANDING_PARTICLES": True}}}
    # Two bodies at separations inside the switching radius, inside the taper band, just inside
    # and beyond the cutoff, in a box whose mesh is wide enough for the reference's cell sweep to
    # visit every cell once (2 * ceil(r_c / cell) + 1 <= MESH_SIZE).  Inputs only: the tests compute
    # the expected accelerations from the written specification,
    #   a_i = S(r) G m_j d [ (r^2 + eps^2)^(-3/2) - (r^2 + eps_PM^2)^(-3/2) ],  eps_PM = 0.5 cell,
    #   S = 2 x^3 - 3 x^2 + 1,  x = (r - r_start) / (r_c - r_start)   (docs/source/book.md:403-455),
    # not from any restatement of src/particles.cpp.
    kat["taper_band_spec"] = {
        "source": "docs/source/book.md:384-455 (specification), src/particles.cpp:321-359 (the code it describes)",
        "config": {"DOMAIN_SIZE": 16.0, "MESH_SIZE": 8, "CELL_SIZE": 2.0, "G": 1.0, "SOFTENING_SQUARED": 0.01,
                   "CUTOFF_RADIUS": 5.0, "CUTOFF_RADIUS_SQUARED": 25.0, "CUTOFF_TRANSITION_WIDTH": 1.0,
                   "R_SWITCH_START": 4.0, "R_SWITCH_START_SQ": 16.0, "USE_PM": True},
        "cases": [
            {"name": "inside the switching radius", "a": [5.0, 6.0, 7.0, 1.5], "b": [7.0, 4.0, 8.0, 0.75]},
            {"name": "middle of the taper band (r = 4.5, S = 0.5)", "a": [5.0, 6.0, 7.0, 1.5],
             "b": [2.0, 7.5, 10.0, 0.75]},
            {"name": "taper band across the periodic boundary (d = 3, -1.5, 2.5 after the wrap)",
             "a": [15.0, 1.0, 8.0, 2.0], "b": [2.0, 15.5, 10.5, 0.25]},
            {"name": "just inside the cutoff (r = 4.96875)", "a": [1.0, 1.0, 1.0, 1.0],
             "b": [5.96875, 1.0, 1.0, 3.0]},
            {"name": "beyond the cutoff (r = 5.03125)", "a": [1.0, 1.0, 1.0, 1.0],
             "b": [1.0, 6.03125, 1.0, 3.0]},
        ]}
    with open(os.path.join(HERE, "reference_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)

    # small systems, long-double sums
    out = {}
    sets = {
        "lattice8": (datasets.jittered_lattice(8), 8, 8),
        "lattice12_even_exact": (datasets.jittered_lattice(12, jitter=0.0), 12, 12),
        "uniform1000": (datasets.uniform_random(1000), 10, 10),
        "clustered1500": (datasets.clustered(1500), 11, 16),
    }
    for name, (pos, n_side_cfg, mesh) in sets.items():
        n = pos[0].size
        m = datasets.masses(n, equal=(name != "clustered1500"))
        for mode in (0, 1):
            p = orc.params_from_config(mode, mesh_size=mesh, n_per_side=n_side_cfg)
            acc = orc.accumulate(p, *pos, m, long_double=True)
            out[f"{name}_acc_mode{mode}"] = np.stack(acc)
            out[f"{name}_params_mode{mode}"] = np.array(
                [p.domain_size, p.G, p.soft_sq, p.pm_soft_sq, p.r_switch, p.r_switch_sq,
                 p.cutoff_sq, p.trans_width, p.cell_size, p.mesh_size, p.use_pm])
        out[f"{name}_pos"] = np.stack(pos)
        out[f"{name}_mass"] = m
    np.savez_compressed(os.path.join(HERE, "pp_small.npz"), **out)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
