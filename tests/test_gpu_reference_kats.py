"""The reference's own gravity-step tests (tests/golden/reference_kat.json, transcribed from
tests/test_gravity.cpp and tests/test_ics.cpp) and the specification-derived taper-band cases,
through the GPU libraries' C ABIs: libnbpp (pair force), libnbpm (mesh) and the C++ host side
(initial conditions).  The CPU oracle runs the same cases in tests/test_reference_kats.py, whose
helpers are reused here."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT
import nbody_amd
from nbody_amd import ffi
from oracle import oracle as O
from test_reference_kats import KAT, approx, pm_plus_pp_setup, spec_acceleration, spec_params

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "nbody_amd", "host", "nbody_host_tests")


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda:0")


@pytest.fixture(scope="module")
def mesh():
    return nbody_amd.load_mesh(device=0)


# ---- the taper band, against the specification ------------------------------------------------
@pytest.mark.parametrize("case", KAT["taper_band_spec"]["cases"], ids=lambda c: c["name"][:28])
@pytest.mark.parametrize("kernel", [ffi.KERNEL_P3M_ALLPAIRS, ffi.KERNEL_P3M_CELLS])
def test_taper_band_two_body_against_the_specification(lib, case, kernel):
    cfg = KAT["taper_band_spec"]["config"]
    a, b = case["a"], case["b"]
    want_a, want_b = spec_acceleration(cfg, a, b)
    pos = [np.array([a[c], b[c]]) for c in range(3)]
    m = np.array([a[3], b[3]])
    p = ffi.Params.from_buffer_copy(bytes(spec_params(cfg)))
    lib.set_option(ffi.OPT_FORCE_KERNEL, kernel)
    try:
        acc = [np.zeros(2) for _ in range(3)]
        lib.accumulate(p, *pos, m, *acc)
        assert lib.stats()["kernel_variant"] == kernel
    finally:
        lib.set_option(ffi.OPT_FORCE_KERNEL, 0)
    scale = max(np.max(np.abs(want_a)), np.max(np.abs(want_b)))
    if "beyond" in case["name"]:
        assert all(np.all(g == 0.0) for g in acc)
        return
    for c in range(3):
        assert abs(acc[c][0] - want_a[c]) <= 1e-13 * scale
        assert abs(acc[c][1] - want_b[c]) <= 1e-13 * scale
    if kernel == ffi.KERNEL_P3M_CELLS:
        assert lib.stats()["pairs_accepted"] == 2


# ---- tests/test_gravity.cpp:13-93, :160-233 through libnbpm ----------------------------------------
def _mesh_params(cfg):
    return ffi.MeshParams(cfg["DOMAIN_SIZE"], cfg.get("G", 1.0), cfg["MESH_SIZE"], 0)


def _deposit(mesh, cfg, particle):
    M = cfg["MESH_SIZE"]
    x, y, z, m = particle
    rho = torch.full((M ** 3,), 7.0, dtype=torch.float64, device="cuda:0")
    mesh.deposit_device(_mesh_params(cfg), dev([x]), dev([y]), dev([z]), dev([m]), rho)
    torch.cuda.synchronize()
    return rho


@pytest.mark.parametrize("case", KAT["cic_deposit"]["cases"], ids=lambda c: c["name"])
def test_cic_deposit(mesh, case):
    cfg = KAT["cic_deposit"]["config"]
    M = cfg["MESH_SIZE"]
    rho = _deposit(mesh, cfg, case["particle"]).cpu().numpy().reshape(M, M, M)
    assert float(np.sum(rho)) * cfg["CELL_VOLUME"] == approx(case["total_mass"])
    for i, j, k, want in case["cells"]:
        assert rho[i, j, k] == approx(want, margin=1e-300 if want else 1e-12)


def _solve(mesh, cfg, rho):
    M = cfg["MESH_SIZE"]
    g = [torch.empty(M ** 3, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    mesh.solve_device(_mesh_params(cfg), rho, *g)
    torch.cuda.synchronize()
    return [c.cpu().numpy().reshape(M, M, M) for c in g]


def test_pm_uniform_density_gives_no_force(mesh):
    cfg, u = KAT["pm_solver"]["config"], KAT["pm_solver"]["uniform"]
    rho = torch.full((cfg["MESH_SIZE"] ** 3,), u["density"], dtype=torch.float64, device="cuda:0")
    for comp in _solve(mesh, cfg, rho):
        assert np.max(np.abs(comp)) <= u["margin"]


def test_pm_point_mass_is_symmetric_and_attractive(mesh):
    cfg, pm = KAT["pm_solver"]["config"], KAT["pm_solver"]["point_mass"]
    rho = _deposit(mesh, cfg, pm["particle"])
    gx = _solve(mesh, cfg, rho)[0]
    assert abs(gx[tuple(pm["cell"])]) <= pm["margin"]
    right, left = gx[tuple(pm["right_neighbour_negative"])], gx[tuple(pm["left_neighbour_positive"])]
    assert right < 0.0 < left
    assert left == approx(-right)


def test_cic_gather_of_a_uniform_field(mesh):
    cfg, k = KAT["cic_gather"]["config"], KAT["cic_gather"]
    M = cfg["MESH_SIZE"]
    x, y, z, _ = k["particle"]
    grids = [torch.full((M ** 3,), v, dtype=torch.float64, device="cuda:0") for v in k["uniform_field"]]
    acc = [torch.full((1,), 9.0, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    mesh.interpolate_device(_mesh_params(cfg), dev([x]), dev([y]), dev([z]), *grids, *acc)
    torch.cuda.synchronize()
    for got, want in zip(acc, k["uniform_field"]):
        assert float(got[0]) == approx(want, margin=1e-15)


def test_pm_plus_pp_is_physically_active(lib, mesh, oracle):
    """tests/test_gravity.cpp:235-269 as the reference runs it: the mesh force overwrites, the pair
    force adds.  The GPU sum is the exact one (every pair once), which the oracle reproduces to
    1e-10 and which satisfies the reference's assertion like the reference's own aliased sum."""
    from oracle import pm_mesh
    k, derived, p, pos, m = pm_plus_pp_setup(oracle)
    M = k["config"]["MESH_SIZE"]
    mp = ffi.MeshParams(1.0, derived["G"], M, 0)
    acc = [np.zeros(2) for _ in range(3)]
    mesh.accelerations(mp, *pos, m, *acc)                     # overwrites (src/integrator.cpp:242-245)
    mesh_only = [a.copy() for a in acc]
    lib.accumulate(ffi.Params.from_buffer_copy(bytes(p)), *pos, m, *acc)    # adds (:254-258)
    assert abs(acc[2][0]) > k["min_abs_acc_z0"] and acc[2][0] > 0.0 > acc[2][1]
    ref_mesh = pm_mesh.initial_mesh_accelerations(pos, m, M, 1.0, derived["G"])
    ref = [a + b for a, b in zip(ref_mesh, oracle.accumulate(p, *pos, m, traversal=O.EXACT))]
    scale = abs(ref[2][0])
    for c in range(3):
        assert np.max(np.abs(acc[c] - ref[c])) <= 1e-10 * scale
        assert np.max(np.abs(mesh_only[c] - ref_mesh[c])) <= 1e-12 * scale


# ---- tests/test_ics.cpp through the C++ initial-condition generator ---------------------------------
def _ics(tmp_path, cfg, name):
    ini = tmp_path / f"{name}.ini"
    ini.write_text(
        f"[domain]\nmesh_size = {cfg['MESH_SIZE']}\nbox_size_mpc = {cfg.get('BOX_SIZE_MPC', 16.0)}\n"
        f"[cosmology]\nomega_baryon = {cfg.get('OMEGA_BARYON', 0.045)}\n"
        f"[initial_conditions]\nn_per_side = {cfg['N_PER_SIDE']}\nseed = {cfg.get('SEED', 42)}\n"
        f"sigma_8 = {cfg.get('SIGMA_8', 0.81)}\n"
        f"standing_particles = {'true' if cfg.get('STANDING_PARTICLES') else 'false'}\n")
    out = tmp_path / f"{name}.bin"
    r = subprocess.run([BIN, "ics", str(ini), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(out, "rb").read()
    n = struct.unpack("q", raw[:8])[0]
    arr = np.frombuffer(raw[8:], dtype=np.float64)
    return arr[: 6 * n].reshape(6, n), arr[6 * n], arr[6 * n + 1], ini


def test_ics_macro_state_and_kinematics(tmp_path, oracle):
    k = KAT["ics"]["macro_state"]
    state, a, H, _ = _ics(tmp_path, k["config"], "macro")
    derived = oracle.config_derive(mesh_size=32, n_per_side=32)
    n = state.shape[1]
    assert abs(float(np.sum(state[3])) * derived["dm_particle_mass"]) < k["dm_momentum_x_below"]
    lo, hi = k["max_abs_vel_x_between"]
    assert lo < float(np.max(np.abs(state[3]))) < hi
    assert n * derived["dm_particle_mass"] + derived["gas_total_mass"] == approx(k["total_mass"],
                                                                                  margin=k["mass_margin"])
    # velocities are H f times the displacement from the lattice (tests/test_ics.cpp:147-201)
    kin = KAT["ics"]["zeldovich_kinematics"]
    state, a, H, _ = _ics(tmp_path, kin["config"], "kin")
    f = (0.3 / (0.3 + 0.7 * a ** 3)) ** 0.55
    spacing = 1.0 / kin["config"]["N_PER_SIDE"]
    for c in range(3):
        moving = np.abs(state[3 + c]) >= 1e-14
        unperturbed = np.fmod(state[c] - state[3 + c] / (H * f) + 1.0, 1.0)
        assert np.max(np.abs(np.fmod(unperturbed, spacing)[moving] - 0.5 * spacing)) <= kin["lattice_margin"]


def test_ics_standing_particles_and_gas_mass(tmp_path, oracle):
    """STANDING_PARTICLES zeroes every initial velocity (tests/test_ics.cpp:271-299); the gas the
    engine starts from carries GAS_TOTAL_MASS (:59-77, :203-239)."""
    k = KAT["ics"]["standing_particles"]
    state, _, _, ini = _ics(tmp_path, k["config"], "standing")
    assert np.all(state[3:6] == 0.0)
    r = subprocess.run([BIN, "run", str(ini), "0"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    gas_mass = float(r.stdout.split("gas mass")[1].split()[0])
    derived = oracle.config_derive(mesh_size=16, n_per_side=16)
    assert gas_mass == approx(derived["gas_total_mass"], margin=KAT["ics"]["macro_state"]["mass_margin"])
