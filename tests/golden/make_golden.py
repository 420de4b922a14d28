"""Regenerates the committed golden vectors.  Run in the build container only:

    python tests/golden/make_golden.py

Sources of truth, in order of authority:
  * reference_config.json, reference_wrap.json: outputs of the reference's OWN compiled
    src/config.cpp and src/math_utils.h (oracle/_ref/libnbref.so, built by oracle/Makefile
    from the files under /root/reference).
  * reference_kat.json: the inputs and expected values written in the reference's own
    tests (tests/test_gravity.cpp:98-155, tests/test_math.cpp:10-48), transcribed as data.
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
