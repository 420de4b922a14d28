"""Everything the reference's own tests hold for the gravity step, run against the CPU oracle
(oracle/pp_oracle.c, oracle/pm_mesh.py, oracle/zeldovich.py, oracle/gas.py).  The data is
tests/golden/reference_kat.json (inputs and expected values transcribed from
tests/test_gravity.cpp, tests/test_ics.cpp); tests/test_gpu_reference_kats.py runs the same
cases through the GPU libraries.

The two-body cases of "taper_band_spec" put pairs inside the switching radius, inside the taper
band and either side of the cutoff, where the reference's own two known answers (1/9, and 0
beyond the cutoff) do not reach.  Their expected values are computed HERE from the written
specification (docs/source/book.md:403-455), in extended precision, independently of
oracle/pp_oracle.c.
"""
import json
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import gas as G
from oracle import oracle as O
from oracle import pm_mesh
from oracle.zeldovich import zeldovich_particles

KAT = json.load(open(os.path.join(GOLDEN, "reference_kat.json")))


def approx(value, rel=1.2e-5, margin=0.0):
    """Catch::Approx: |got - value| <= max(margin, epsilon * |value|) (default epsilon ~1.2e-5)."""
    return pytest.approx(value, rel=rel, abs=margin if margin else 1e-300)


# ---- specification of the short-range pair force (book.md:403-455) -------------------------
def spec_acceleration(cfg, a, b):
    """Acceleration of body `a` due to body `b` and of `b` due to `a`, as the specification
    writes it, in long double."""
    ld = np.longdouble
    L = ld(cfg["DOMAIN_SIZE"])
    d = np.array([ld(b[c]) - ld(a[c]) for c in range(3)], dtype=ld)
    d = np.where(d > L / 2, d - L, d)
    d = np.where(d < -L / 2, d + L, d)                       # the shortest periodic separation
    r2 = np.sum(d * d)
    r = np.sqrt(r2)
    rc, rs = ld(cfg["CUTOFF_RADIUS"]), ld(cfg["R_SWITCH_START"])
    if r >= rc:
        S = ld(0)
    elif r <= rs:
        S = ld(1)
    else:
        x = (r - rs) / (rc - rs)
        S = 2 * x ** 3 - 3 * x ** 2 + 1
    eps2 = ld(cfg["SOFTENING_SQUARED"])
    eps_pm2 = (ld(0.5) * ld(cfg["CELL_SIZE"])) ** 2
    shape = (r2 + eps2) ** ld(-1.5) - (r2 + eps_pm2) ** ld(-1.5)
    Gc = ld(cfg["G"])
    return (S * Gc * ld(b[3]) * shape * d).astype(np.float64), (-S * Gc * ld(a[3]) * shape * d).astype(np.float64)


def spec_params(cfg):
    return O.Params(cfg["DOMAIN_SIZE"], cfg["G"], cfg["SOFTENING_SQUARED"], (0.5 * cfg["CELL_SIZE"]) ** 2,
                    cfg["R_SWITCH_START"], cfg["R_SWITCH_START_SQ"], cfg["CUTOFF_RADIUS_SQUARED"],
                    cfg["CUTOFF_TRANSITION_WIDTH"], cfg["CELL_SIZE"], cfg["MESH_SIZE"], 1)


@pytest.mark.parametrize("case", KAT["taper_band_spec"]["cases"], ids=lambda c: c["name"][:28])
@pytest.mark.parametrize("traversal", [O.EXACT, O.REF_OFFLOAD, O.REF_CPU, O.EXACT_CELLS])
def test_taper_band_two_body_against_the_specification(oracle, case, traversal):
    cfg = KAT["taper_band_spec"]["config"]
    a, b = case["a"], case["b"]
    want_a, want_b = spec_acceleration(cfg, a, b)
    pos = [np.array([a[c], b[c]]) for c in range(3)]
    m = np.array([a[3], b[3]])
    got = oracle.accumulate(spec_params(cfg), *pos, m, traversal=traversal)
    scale = max(np.max(np.abs(want_a)), np.max(np.abs(want_b)))
    if "beyond" in case["name"]:
        assert scale == 0.0 and all(np.all(g == 0.0) for g in got)
        return
    assert scale > 0.0
    for c in range(3):
        assert abs(got[c][0] - want_a[c]) <= 1e-14 * scale
        assert abs(got[c][1] - want_b[c]) <= 1e-14 * scale


def test_taper_cases_cover_the_band():
    """The cases are where they claim to be: S = 1, 0 < S < 1 (one of them exactly 1/2), S = 0."""
    cfg = KAT["taper_band_spec"]["config"]
    seen = set()
    for case in KAT["taper_band_spec"]["cases"]:
        d = np.array(case["b"][:3]) - np.array(case["a"][:3])
        d -= cfg["DOMAIN_SIZE"] * np.round(d / cfg["DOMAIN_SIZE"])
        r = float(np.linalg.norm(d))
        seen.add("inside" if r <= cfg["R_SWITCH_START"] else "band" if r < cfg["CUTOFF_RADIUS"] else "beyond")
        if "S = 0.5" in case["name"]:
            assert r == 4.5
    assert seen == {"inside", "band", "beyond"}
    # the reference's cell sweep visits every cell once for this mesh (no aliasing)
    assert 2 * math.ceil(cfg["CUTOFF_RADIUS"] / cfg["CELL_SIZE"]) + 1 <= cfg["MESH_SIZE"]


# ---- tests/test_gravity.cpp:13-93: CIC density assignment ---------------------------------------
@pytest.mark.parametrize("case", KAT["cic_deposit"]["cases"], ids=lambda c: c["name"])
def test_cic_deposit(case):
    cfg = KAT["cic_deposit"]["config"]
    x, y, z, m = case["particle"]
    rho, _, _ = pm_mesh.cic_deposit([np.array([x]), np.array([y]), np.array([z])], np.array([m]),
                                    cfg["MESH_SIZE"], cfg["DOMAIN_SIZE"])
    assert float(np.sum(rho)) * cfg["CELL_VOLUME"] == approx(case["total_mass"])
    for i, j, k, want in case["cells"]:
        assert rho[i, j, k] == approx(want, margin=1e-300 if want else 1e-12)


# ---- tests/test_gravity.cpp:160-206: Poisson solve ------------------------------------------------
def test_pm_uniform_density_gives_no_force():
    cfg, u = KAT["pm_solver"]["config"], KAT["pm_solver"]["uniform"]
    M = cfg["MESH_SIZE"]
    g = pm_mesh.poisson_accelerations(np.full((M, M, M), u["density"]), cfg["DOMAIN_SIZE"], cfg["G"])
    for comp in g:
        assert np.max(np.abs(comp)) <= u["margin"]


def test_pm_point_mass_is_symmetric_and_attractive():
    cfg, pm = KAT["pm_solver"]["config"], KAT["pm_solver"]["point_mass"]
    x, y, z, m = pm["particle"]
    rho, _, _ = pm_mesh.cic_deposit([np.array([x]), np.array([y]), np.array([z])], np.array([m]),
                                    cfg["MESH_SIZE"], cfg["DOMAIN_SIZE"])
    assert rho[tuple(pm["cell"])] * cfg["CELL_SIZE"] ** 3 == approx(m)     # all of it in one cell
    gx = pm_mesh.poisson_accelerations(rho, cfg["DOMAIN_SIZE"], cfg["G"])[0]
    assert abs(gx[tuple(pm["cell"])]) <= pm["margin"]
    right, left = gx[tuple(pm["right_neighbour_negative"])], gx[tuple(pm["left_neighbour_positive"])]
    assert right < 0.0 < left
    assert left == approx(-right)


# ---- tests/test_gravity.cpp:208-233: CIC interpolation ---------------------------------------------
def test_cic_gather_of_a_uniform_field():
    cfg, k = KAT["cic_gather"]["config"], KAT["cic_gather"]
    M = cfg["MESH_SIZE"]
    x, y, z, m = k["particle"]
    _, idx, weights = pm_mesh.cic_deposit([np.array([x]), np.array([y]), np.array([z])], np.array([m]),
                                          M, cfg["DOMAIN_SIZE"])
    grids = [np.full((M, M, M), v) for v in k["uniform_field"]]
    acc = pm_mesh.cic_interpolate(grids, idx, weights)
    for got, want in zip(acc, k["uniform_field"]):
        assert got[0] == approx(want, margin=1e-15)


# ---- tests/test_gravity.cpp:235-269: mesh + short range together -------------------------------------
def pm_plus_pp_setup(oracle):
    k = KAT["pm_plus_pp_active"]
    derived = oracle.config_derive(mesh_size=k["config"]["MESH_SIZE"], n_per_side=k["config"]["N_PER_SIDE"])
    p = oracle.params_from_config(True, mesh_size=k["config"]["MESH_SIZE"], n_per_side=k["config"]["N_PER_SIDE"])
    pos = [np.array([q[c] for q in k["particles"]]) for c in range(3)]
    m = np.array([q[3] for q in k["particles"]])
    return k, derived, p, pos, m


def test_pm_plus_pp_is_physically_active(oracle):
    """The reference's only test with a non-zero short-range term.  Its mesh of 4 cells is narrower
    than the 7 cells the sweep visits per axis, so the reference (REF_CPU traversal) meets the one
    pair twice (cell offsets +1 and -3 along z are the same cell); the assertion it makes,
    |acc_z[0]| > 0.05, holds for that sum and for the exact one."""
    k, derived, p, pos, m = pm_plus_pp_setup(oracle)
    M = k["config"]["MESH_SIZE"]
    mesh = pm_mesh.initial_mesh_accelerations(pos, m, M, 1.0, derived["G"])
    exact = oracle.accumulate(p, *pos, m, traversal=O.EXACT)
    as_reference = oracle.accumulate(p, *pos, m, traversal=O.REF_CPU)
    assert exact[2][0] > 0.0 > exact[2][1]                       # attraction along z
    assert as_reference[2][0] == pytest.approx(2.0 * exact[2][0], rel=1e-15)
    for short in (exact, as_reference):
        total_z0 = mesh[2][0] + short[2][0]
        assert abs(total_z0) > k["min_abs_acc_z0"] and total_z0 > 0.0
    assert mesh[0][0] == pytest.approx(0.0, abs=1e-15) and exact[0][0] == 0.0 and exact[1][0] == 0.0


# ---- tests/test_ics.cpp ---------------------------------------------------------------------------------
def test_internal_energy_units():
    k = KAT["ics"]["internal_energy"]
    u = G.internal_energy_from_temperature(k["T_kelvin"], k["hubble_param"], k["box_size_mpc"], k["domain_size"],
                                           k["gamma"])
    assert u == approx(k["u_code"], margin=k["margin"])
    u2 = G.internal_energy_from_temperature(k["T_kelvin"], k["hubble_param"], k["box_size_mpc"], 2.0, k["gamma"])
    assert u2 == approx(u * k["double_domain_factor"])


def _initial_state(oracle, cfg, standing=False):
    M, N = cfg["MESH_SIZE"], cfg["N_PER_SIDE"]
    ob = cfg.get("OMEGA_BARYON", 0.045)
    derived = oracle.config_derive(mesh_size=M, n_per_side=N, omega_baryon=ob)
    pos, vel, info = zeldovich_particles(n_per_side=N, mesh_size=M, seed=cfg.get("SEED", 42),
                                         box_size_mpc=cfg.get("BOX_SIZE_MPC", 16.0),
                                         sigma_8=cfg.get("SIGMA_8", 0.81), oracle=oracle)
    if standing:                                                  # src/ics.cpp:200-208
        vel = [np.zeros_like(v) for v in vel]
    mass = np.full(N ** 3, derived["dm_particle_mass"])
    rho, _, _ = pm_mesh.cic_deposit(pos, mass, M, 1.0)
    gas = G.initialize_gas(rho, info["displacement"], info["growth_f"], info["hubble_param"],
                           derived["gas_total_mass"], N ** 3 * derived["dm_particle_mass"], 5.0 / 3.0,
                           box_size_mpc=cfg.get("BOX_SIZE_MPC", 16.0), standing=standing)
    return derived, pos, vel, mass, rho, gas, info


def test_ics_macro_state(oracle):
    k = KAT["ics"]["macro_state"]
    derived, pos, vel, mass, rho, gas, _ = _initial_state(oracle, k["config"])
    M = k["config"]["MESH_SIZE"]
    V = (1.0 / M) ** 3
    gas_mass = float(np.sum(gas.density)) * V
    assert float(np.sum(mass)) + gas_mass == approx(k["total_mass"], margin=k["mass_margin"])
    assert gas_mass == approx(derived["gas_total_mass"], margin=k["mass_margin"])
    assert abs(float(np.sum(vel[0] * mass))) < k["dm_momentum_x_below"]
    assert abs(float(np.sum(gas.momentum_x)) * V) < k["gas_momentum_x_below"]
    lo, hi = k["max_abs_vel_x_between"]
    assert lo < float(np.max(np.abs(vel[0]))) < hi


def test_zeldovich_field_is_linear_in_sigma8(oracle):
    k = KAT["ics"]["sigma8_linearity"]
    c = k["config"]
    fields = [zeldovich_particles(n_per_side=4, mesh_size=c["MESH_SIZE"], seed=c["SEED"],
                                  box_size_mpc=c["BOX_SIZE_MPC"], sigma_8=s8, spectral_index=c["SPECTRAL_INDEX"],
                                  start_a=k["start_a"], oracle=oracle)[2]["displacement"] for s8 in k["sigma_8"]]
    for base, double in zip(*fields):
        assert np.max(np.abs(double - k["factor"] * base)) <= k["margin"]
        assert np.max(np.abs(base)) > 1e-5                        # not a comparison of zeros


def test_zeldovich_kinematics(oracle):
    k = KAT["ics"]["zeldovich_kinematics"]
    derived, pos, vel, _, _, _, info = _initial_state(oracle, k["config"])
    N = k["config"]["N_PER_SIDE"]
    spacing = 1.0 / N
    ratio = info["hubble_param"] * info["growth_f"]
    for p, v in zip(pos, vel):
        moving = np.abs(v) >= 1e-14
        unperturbed = np.fmod(p - v / ratio + 1.0, 1.0)
        assert np.max(np.abs(np.fmod(unperturbed, spacing)[moving] - 0.5 * spacing)) <= k["lattice_margin"]


def test_gas_density_traces_dark_matter(oracle):
    k = KAT["ics"]["gas_traces_dm"]
    derived, _, _, mass, rho, gas, _ = _initial_state(oracle, k["config"])
    expected = rho * (derived["gas_total_mass"] / float(np.sum(mass)))
    floored = expected < k["floor"]
    assert np.max(np.abs(gas.density - expected)[~floored], initial=0.0) <= k["margin"]
    assert np.all(gas.density[floored] == k["floor"])


def test_zeldovich_field_has_zero_mean(oracle):
    k = KAT["ics"]["zero_mean"]
    c = k["config"]
    disp = zeldovich_particles(n_per_side=4, mesh_size=c["MESH_SIZE"], seed=c["SEED"], box_size_mpc=c["BOX_SIZE_MPC"],
                               sigma_8=c["SIGMA_8"], start_a=k["start_a"], oracle=oracle)[2]["displacement"]
    for d in disp:
        assert abs(float(np.sum(d)) / d.size) < k["mean_below"]


def test_standing_particles_start_at_rest(oracle):
    k = KAT["ics"]["standing_particles"]
    _, _, vel, _, _, gas, _ = _initial_state(oracle, k["config"], standing=True)
    assert all(np.all(v == 0.0) for v in vel)
    assert all(np.all(mom == 0.0) for mom in (gas.momentum_x, gas.momentum_y, gas.momentum_z))
