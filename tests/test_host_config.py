"""The Python mirror of Config against the reference's compiled config.cpp outputs."""
import json
import os

from conftest import GOLDEN
from nbody_amd import Config


def test_config_mirror_matches_reference_build():
    gold = json.load(open(os.path.join(GOLDEN, "reference_config.json")))
    for name, ref in gold.items():
        c = Config(DOMAIN_SIZE=ref["domain_size"], MESH_SIZE=ref["mesh_size"],
                   N_PER_SIDE=ref["n_per_side"], OMEGA_BARYON=ref["omega_baryon"],
                   OMEGA_M=ref["omega_m"], CUTOFF_RADIUS_CELLS=ref["cutoff_radius_cells"],
                   CUTOFF_TRANSITION_WIDTH_FACTOR=ref["cutoff_transition_width_factor"],
                   DT_FACTOR=ref["dt_factor"])
        for k in ("cell_size", "cell_volume", "omega_dm", "cutoff_radius",
                  "cutoff_radius_squared", "cutoff_transition_width", "r_switch_start",
                  "r_switch_start_sq", "num_dm_particles", "G", "dm_particle_mass",
                  "gas_total_mass", "softening_squared", "fixed_dt"):
            assert getattr(c, k.upper() if k != "G" else "G") == ref[k], (name, k)


def test_pp_params_hoist():
    c = Config()
    p = c.pp_params()
    assert p.pm_soft_sq == (0.5 * c.CELL_SIZE) * (0.5 * c.CELL_SIZE)
    assert p.use_pm == 1 and p.mesh_size == 32 and p.cutoff_sq == c.CUTOFF_RADIUS_SQUARED
