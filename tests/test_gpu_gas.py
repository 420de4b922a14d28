"""GPU parity of libnbgas (the gas side of the step, through its C ABI) against the numpy
restatement in oracle/gas.py, and of the whole step (libnbpp + libnbpm + libnbgas) against the
accelerations a build of the reference printed.

Every kernel keeps the reference's operation order and IEEE division / square root, so meshes are
compared to 1e-13 of the field's largest magnitude (observed: equal or within an ulp); the CFL
time step, a max reduction, is compared exactly.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from nbody_amd import Config, ffi
from oracle import gas as G
from oracle import full_step as F
from oracle import pm_mesh

pytestmark = pytest.mark.gpu
TOL = 1e-13


@pytest.fixture(scope="module")
def gaslib():
    return ffi.load_gas(device=0)


@pytest.fixture(scope="module")
def mesh():
    return ffi.load_mesh(device=0)


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0").reshape(-1)


def upload(g):
    return {f: dev(getattr(g, f)) for f in G.Gas.FIELDS}


def params(M, cell, gamma=5.0 / 3.0, cfl=0.5):
    p = ffi.GasParams()
    p.cell_size, p.gamma, p.cfl_safety_factor, p.mesh_size = cell, gamma, cfl, M
    return p


def assert_state_close(t, g, fields=G.Gas.FIELDS, tol=TOL):
    torch.cuda.synchronize()
    for f in fields:
        got = t[f].cpu().numpy().reshape(getattr(g, f).shape)
        ref = getattr(g, f)
        scale = max(float(np.max(np.abs(ref))), 1e-300)
        assert float(np.max(np.abs(got - ref))) <= tol * scale, f


def random_gas(M, seed, with_vacuum=True, speed=0.3):
    rng = np.random.default_rng(seed)
    g = G.Gas(M)
    g.density = 0.5 + rng.random((M, M, M))
    v = [speed * rng.standard_normal((M, M, M)) for _ in range(3)]
    g.momentum_x, g.momentum_y, g.momentum_z = (g.density * c for c in v)
    g.energy = 0.4 * rng.random((M, M, M)) + 0.5 * g.density * sum(c * c for c in v)
    if with_vacuum:
        g.density[0, 1, 2] = 1e-13           # vacuum cell
        g.density[3, 3, 3] = 1e-12           # exactly the floor: still vacuum (strict >)
        g.energy[1, 1, 1] = 0.0              # pressure floor
    return g


@pytest.mark.parametrize("M,gamma", [(8, 5.0 / 3.0), (12, 1.4), (5, 1.6666666667)])
def test_primitives_match_oracle(gaslib, M, gamma):
    g = random_gas(M, M)
    t = upload(g)
    gaslib.update_primitives_device(params(M, 1.0 / M, gamma), t)
    G.update_primitive_variables(g, gamma)
    assert_state_close(t, g, tol=0.0)        # one division, products and sums: exact


@pytest.mark.parametrize("M,gamma,dt,speed", [(8, 5.0 / 3.0, 0.004, 0.3), (16, 1.4, 0.002, 1.5),
                                               (7, 1.6666666667, 0.01, 0.05), (32, 5.0 / 3.0, 0.001, 3.0)])
def test_hydro_step_matches_oracle(gaslib, M, gamma, dt, speed):
    """Subsonic, transonic and supersonic mixes (all three branches of the HLL flux), vacuum
    cells and floored pressures, odd and even meshes."""
    g = random_gas(M, 10 + M, speed=speed)
    G.update_primitive_variables(g, gamma)
    t = upload(g)
    cell = 1.0 / M
    gaslib.hydro_step_device(params(M, cell, gamma), t, dt)
    G.hydro_step(g, dt, cell, gamma)
    assert_state_close(t, g)


def test_three_steps_in_a_row_and_cfl(gaslib):
    M, gamma = 16, 5.0 / 3.0
    g = random_gas(M, 3, with_vacuum=False, speed=0.8)
    G.update_primitive_variables(g, gamma)
    t = upload(g)
    p = params(M, 1.0 / M, gamma, cfl=0.4)
    for _ in range(3):
        dt = G.cfl_timestep(g, 1.0 / M, gamma, 0.4)
        assert gaslib.cfl_timestep_device(p, t) == pytest.approx(dt, rel=1e-13)
        gaslib.cfl_timestep_enqueue_device(p, t)          # the same in two halves
        assert gaslib.cfl_timestep_finish(p) == gaslib.cfl_timestep_device(p, t)
        G.hydro_step(g, dt, 1.0 / M, gamma)
        gaslib.hydro_step_device(p, t, dt)
    assert_state_close(t, g, tol=1e-12)
    mass = float(t["density"].sum().cpu())
    assert mass == pytest.approx(float(np.sum(g.density)), rel=1e-13)


def test_cfl_of_gas_at_rest_without_pressure_is_infinite(gaslib):
    M = 4
    g = G.Gas(M)                              # all zero: c_s is NaN -> 0, |v| = 0
    t = upload(g)
    assert gaslib.cfl_timestep_device(params(M, 0.25), t) == np.inf
    assert G.cfl_timestep(g, 0.25, 5.0 / 3.0, 0.5) == np.inf


@pytest.mark.parametrize("a,H", [(1.0, 0.0), (0.02, 235.70446027301335)])
def test_gravity_kick_matches_oracle(gaslib, a, H):
    M, gamma = 12, 5.0 / 3.0
    rng = np.random.default_rng(5)
    g = random_gas(M, 7)
    grav = [rng.standard_normal((M, M, M)) for _ in range(3)]
    t = upload(g)
    gaslib.kick_device(params(M, 1.0 / M, gamma), t, [dev(x) for x in grav], 0.003, a, H)
    G.apply_gas_kick(g, grav, 0.003, a, H, gamma)
    assert_state_close(t, g, tol=1e-15)


@pytest.mark.parametrize("M", [12, 40, 168])
def test_kick_with_cfl_in_one_pass_equals_the_two_calls(gaslib, M):
    """nbgas_kick_cfl_enqueue_device against nbgas_kick_device + nbgas_cfl_timestep_device: the same
    nine meshes and the same time step, bit for bit (vacuum cells and floored pressures included).
    168^3 cells are more workgroups than the read-back buffer has room for partial maxima: the
    extra reduction pass."""
    gamma = 5.0 / 3.0
    rng = np.random.default_rng(M)
    g = random_gas(M, 20 + M, speed=0.7)
    grav = [dev(rng.standard_normal((M, M, M))) for _ in range(3)]
    p = params(M, 1.0 / M, gamma, cfl=0.45)
    two, one = upload(g), upload(g)
    gaslib.kick_device(p, two, grav, 0.003, 0.4, 31.0)
    dt_two = gaslib.cfl_timestep_device(p, two)
    gaslib.kick_cfl_enqueue_device(p, one, grav, 0.003, 0.4, 31.0)
    dt_one = gaslib.cfl_timestep_finish(p)
    torch.cuda.synchronize()
    assert dt_one == dt_two and np.isfinite(dt_one)
    for f in G.Gas.FIELDS:
        assert torch.equal(one[f], two[f]), f


def test_total_density_and_initial_state(gaslib, oracle):
    """initialize_gas on the Zel'dovich field of the shipped configuration."""
    from oracle.zeldovich import zeldovich_particles
    M = 32
    pos, _, info = zeldovich_particles(oracle=oracle)
    cfg = json.load(open(os.path.join(GOLDEN, "reference_config.json")))["simulation_ini"]
    m = np.full(M ** 3, cfg["dm_particle_mass"])
    rho, _, _ = pm_mesh.cic_deposit(pos, m, M, 1.0)
    gamma = 1.6666666667
    ref = G.initialize_gas(rho, info["displacement"], info["growth_f"], info["hubble_param"],
                           cfg["gas_total_mass"], M ** 3 * cfg["dm_particle_mass"], gamma)
    t = {f: torch.zeros(M ** 3, dtype=torch.float64, device="cuda:0") for f in G.Gas.FIELDS}
    u0 = G.internal_energy_from_temperature(50.0, 0.7, 16.0, 1.0, gamma)
    p = params(M, 1.0 / M, gamma)
    gaslib.initialize_device(p, t, dev(rho), [dev(d) for d in info["displacement"]],
                             cfg["gas_total_mass"] / (M ** 3 * cfg["dm_particle_mass"]),
                             info["hubble_param"], info["growth_f"], u0)
    assert_state_close(t, ref, tol=1e-15)
    total = torch.zeros(M ** 3, dtype=torch.float64, device="cuda:0")
    gaslib.total_density_device(p, t, dev(rho), total)
    torch.cuda.synchronize()
    assert np.array_equal(total.cpu().numpy().reshape(M, M, M), rho + ref.density)


def test_bad_arguments(gaslib):
    M = 4
    t = upload(G.Gas(M))
    bad = params(M, 0.25, gamma=1.0)
    with pytest.raises(ffi.NbppError):
        gaslib.hydro_step_device(bad, t, 0.1)
    with pytest.raises(ffi.NbppError):
        gaslib.hydro_step_device(params(1, 0.25), t, 0.1)
    with pytest.raises(ffi.NbppError):
        gaslib.hydro_step_device(params(M, 0.25), t, float("nan"))


def test_whole_step_on_the_gpu_matches_reference_output(lib, mesh, gaslib, oracle):
    """initialize_state and one KDK_step of the shipped simulation.ini with every force and the
    gas on the GPU (the three libraries), against the accelerations the reference printed
    (tests/golden/reference_kat.json) and against the whole-step oracle."""
    from oracle.zeldovich import zeldovich_particles
    kat = json.load(open(os.path.join(GOLDEN, "reference_kat.json")))["default_ini_anchor"]
    cfg = json.load(open(os.path.join(GOLDEN, "reference_config.json")))["simulation_ini"]
    M = n_side = 32
    n = n_side ** 3
    gamma = 1.6666666667
    config = Config(OMEGA_BARYON=cfg["omega_baryon"], GAMMA=gamma)
    pp, mp, gp = config.pp_params(), config.mesh_params(), config.gas_params()
    pos, vel, info = zeldovich_particles(oracle=oracle)
    x, v = [dev(a) for a in pos], [dev(a) for a in vel]
    m = dev(np.full(n, config.DM_PARTICLE_MASS))
    acc = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    rho, total = (torch.zeros(M ** 3, dtype=torch.float64, device="cuda:0") for _ in range(2))
    grav = [torch.zeros(M ** 3, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    gas = {f: torch.zeros(M ** 3, dtype=torch.float64, device="cuda:0") for f in G.Gas.FIELDS}

    def forces():                                                    # src/integrator.cpp:236-258
        mesh.deposit_device(mp, *x, m, rho)
        gaslib.total_density_device(gp, gas, rho, total)
        mesh.solve_device(mp, total, *grav)
        mesh.interpolate_device(mp, *x, *grav, *acc)
        lib.accumulate_device(pp, *x, m, *acc)

    a0, H0 = info["scale_factor"], info["hubble_param"]
    mesh.deposit_device(mp, *x, m, rho)                              # src/ics.cpp:299-303
    gaslib.initialize_device(gp, gas, rho, [dev(d) for d in info["displacement"]],
                             config.GAS_TOTAL_MASS / (n * config.DM_PARTICLE_MASS), H0, info["growth_f"],
                             G.internal_energy_from_temperature(50.0, 0.7, 16.0, 1.0, gamma))
    forces()
    torch.cuda.synchronize()
    for c in range(3):
        assert float(acc[c][0]) == pytest.approx(kat["p0_acc_initial_pm_plus_pp"][c], rel=1e-12)

    dt = config.FIXED_DT                                             # src/integrator.cpp:215-275
    gaslib.kick_device(gp, gas, grav, dt / 2.0, a0, H0)
    lib.kick_device(v, acc, dt / 2.0, a0, H0)
    lib.drift_device(x, v, dt, config.DOMAIN_SIZE)
    gaslib.hydro_step_device(gp, gas, dt)
    a1, H1 = F.cosmology(dt)
    forces()
    gaslib.kick_device(gp, gas, grav, dt / 2.0, a1, H1)
    lib.kick_device(v, acc, dt / 2.0, a1, H1)
    torch.cuda.synchronize()
    for c in range(3):
        assert float(acc[c][0]) == pytest.approx(kat["p0_acc_after_one_step"][c], rel=1e-11)

    # and the complete state against the whole-step oracle
    s = F.Settings(oracle, omega_baryon=cfg["omega_baryon"], gamma=gamma, gravity_dt_factor=0.2)
    st = F.initialize_state(s)
    F.kdk_step(s, st, dt)
    assert_state_close(gas, st.gas, tol=1e-11)
    for c in range(3):
        assert np.max(np.abs(x[c].cpu().numpy() - st.pos[c])) < 1e-14
        ref_v = st.vel[c]
        assert np.max(np.abs(v[c].cpu().numpy() - ref_v)) < 1e-11 * np.max(np.abs(ref_v))
    assert gaslib.cfl_timestep_device(gp, gas) == pytest.approx(
        G.cfl_timestep(st.gas, config.CELL_SIZE, gamma, 0.5), rel=1e-11)
